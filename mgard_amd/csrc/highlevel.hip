// High-level path of libmgard_hip.so: whole-array compress / decompress (SURVEY.md section 8b
// "high-level", 8f ranks 2-4) on top of the low-level C ABI of capi.hip:
//   domain decomposition + double-buffered subdomain pipeline
//     (reference DomainDecomposer.hpp:72-470, 649-845; GPUPipelines.hpp:69-207, 330-520),
//   Huffman [+ Zstd] lossless stage (huffman.hpp; Lossless.hpp:70-118, Zstd.hpp:69-128),
//   self-describing container (format.hpp; Metadata.cpp:249-462).
// Host code is C++; everything exported is extern "C" (include/mgard_hip_compress.h).
#include "../../include/mgard_hip_compress.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <memory>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <condition_variable>
#include <thread>
#include <vector>

#include "format.hpp"
#include "huffman.hpp"
#include "env.hpp"

extern "C" void mgh_set_last_error_(const char *msg);  // capi.hip

namespace {
constexpr int kOutlierOverflow = -1000;  // internal: more outliers than the buffers hold

// Kernel attributes such as the dynamic-LDS limit belong to the function on the current device and
// are set once per device ordinal: `if (hl_attr_pending(once)) { set ...; hl_attr_done(once); }`.
// The bit is set AFTER the attributes exist, so a second host thread on the same device can never
// launch before them (two threads setting them twice is harmless).
inline uint64_t hl_device_bit() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return (uint64_t)1 << (dev & 63);
}
inline bool hl_attr_pending(const std::atomic<uint64_t> &done) {
  return !(done.load(std::memory_order_acquire) & hl_device_bit());
}
inline void hl_attr_done(std::atomic<uint64_t> &done) {
  done.fetch_or(hl_device_bit(), std::memory_order_release);
}
// compute units of the current device (cached per device ordinal)
inline int hl_num_cu() {
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v <= 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}


using namespace mgh;

// MGH_DEBUG_SYNC=1: name the stages on stderr and synchronise (developer aid, see capi.hip)
inline void hl_debug(const char *what) {
  static const bool on = env_get("MGH_DEBUG_SYNC", 0) != 0;
  static const bool timing = env_get("MGH_HL_TIMING", 0) != 0;  // + microseconds since the last mark
  if (on || timing) {
    (void)hipDeviceSynchronize();
    static auto last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    if (timing)
      std::fprintf(stderr, "[mgh-hl] %8.1f us  %s\n",
                   std::chrono::duration<double, std::micro>(now - last).count(), what);
    else
      std::fprintf(stderr, "[mgh-hl] %s\n", what);
    std::fflush(stderr);
    last = std::chrono::steady_clock::now();
  }
}

int hl_fail(int code, const std::string &msg) {
  mgh_set_last_error_(msg.c_str());
  return code;
}

#define HL_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return hl_fail(_e == hipErrorOutOfMemory ? MGH_ERR_OUT_OF_MEMORY : MGH_ERR_DEVICE,   \
                     std::string(#expr) + ": " + hipGetErrorString(_e));                   \
  } while (0)
#define HL_TRY(expr)                       \
  do {                                     \
    int _rc = (expr);                      \
    if (_rc != MGH_SUCCESS) return _rc;    \
  } while (0)

// MemoryManager::IsDevicePointer
bool is_device_pointer(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice;
}

// device memory OF DEVICE `dev`: the zero-copy paths (encoder writing into the caller's record,
// decoders reading the code units in place) let KERNELS touch the caller's buffer, which works
// only on the device the kernels run on; memory of another GPU goes through hipMemcpyAsync
bool is_device_pointer_on(const void *p, int dev) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice && a.device == dev;
}

bool is_registered_host(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// grow-only device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  int ensure(size_t bytes) {
    if (bytes <= cap) return MGH_SUCCESS;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HL_HIP(hipMalloc(&p, bytes));
    cap = bytes;
    return MGH_SUCCESS;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// grow-only pinned host buffer: the small host<->device transfers of the lossless stage (histogram,
// code table, record head, counts) go through it -- out of pageable memory every one of them is a
// staged, host-synchronous copy of 15-25 us (rocprofv3 timeline of mgh_compress: ten of them in a row)
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return MGH_SUCCESS;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    HL_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return MGH_SUCCESS;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// ---- Zstd through the system library (libzstd.so.1), resolved at first use -----------------
struct ZstdApi {
  void *lib = nullptr;
  size_t (*compressBound)(size_t) = nullptr;
  size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
  size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  bool tried = false;
  bool load() {
    if (tried) return lib != nullptr;
    tried = true;
    for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    compressBound = (size_t(*)(size_t))dlsym(lib, "ZSTD_compressBound");
    compress = (size_t(*)(void *, size_t, const void *, size_t, int))dlsym(lib, "ZSTD_compress");
    decompress = (size_t(*)(void *, size_t, const void *, size_t))dlsym(lib, "ZSTD_decompress");
    isError = (unsigned (*)(size_t))dlsym(lib, "ZSTD_isError");
    if (!compressBound || !compress || !decompress || !isError) {
      dlclose(lib);
      lib = nullptr;
    }
    return lib != nullptr;
  }
};
ZstdApi g_zstd;

// ---- serialized Huffman record: offsets with natural alignment (Huffman.hpp:163-239) --------
inline size_t align_up(size_t off, size_t a) { return (off + a - 1) / a * a; }

struct PayloadLayout {
  size_t primary_count = 0, huffmeta = 0, decodebook_size = 0, decodebook = 0, ddata_size = 0,
         ddata = 0, outlier_count = 0, outlier_idx = 0, outliers = 0, total = 0;
  // Behind the reference's payload (its reader stops at the outlier lists), optional: the decoder's
  // synchronisation points (huffman.hpp: k_encode_chain), [u64 kSyncTag][u32 x 64 per chunk]. The
  // section's size is 8 (mod 16) while the outlier lists' is 0: a reader that knows the record's
  // size and where the lists start sees from the remainder whether it is there.
  size_t sync_tag = 0, sync = 0;
  static constexpr uint64_t kSyncTag = 0x31434e595348474dull;  // "MGHSYNC1"
  static size_t sync_bytes(size_t nchunk) { return 8 + 4 * (size_t)huff::kSyncLanes * nchunk; }
  // offsets of: primary_count, dict_size, chunk_size, huffmeta_size are fixed (0, 8, 12, 16)
  void compute(size_t nchunk, size_t dict, size_t units, size_t noutlier, bool with_sync = false) {
    size_t off = 0;
    primary_count = off; off += 8;
    off += 4;  // dict_size (int)
    off += 4;  // chunk_size (int)
    off = align_up(off, 8); off += 8;  // huffmeta_size
    huffmeta = off; off += 8 * 2 * nchunk;
    decodebook_size = off; off += 8;
    decodebook = off; off += 8 * (2 * 64) + 8 * dict;
    off = align_up(off, 8);
    ddata_size = off; off += 8;
    off = align_up(off, 8);
    ddata = off; off += 8 * units;
    outlier_count = off; off += 8;
    outlier_idx = off; off += 8 * noutlier;
    outliers = off; off += 8 * noutlier;
    sync_tag = sync = 0;
    if (with_sync) {
      sync_tag = off; off += 8;
      sync = off; off += 4 * (size_t)huff::kSyncLanes * nchunk;
    }
    total = off;
  }
};

} // namespace

// ---- lossless context ----------------------------------------------------------------------
struct mgh_lossless_ctx {
  int dev = 0;
  DevBuf freq, code, bits, entry, total, units, tables, oidx, oval, state, dtable, sync;
  bool use_sync = false;   // the record of the last compress call carries synchronisation points
  uint64_t prepared_n = 0, prepared_dict = 0, prepared_chunk = 0;  // lossless_prepare() ran for a record of this size
  size_t n_chunks = 0;
  bool overflow = false;  // the code stream did not fit into cap_units: treat as incompressible
  huff::Codebook codebook;
  std::vector<uint8_t> host;   // serialized payload (when assembled on the host)
  std::vector<uint8_t> host2;  // zstd scratch
  // the record of the last compress call in pieces: everything before the code units sits in
  // `head`, the units and the outlier lists are still on the device
  std::vector<uint8_t> head;   // (decompression: host copy of the leading part of a record)
  // compression: [8 bytes: the record's size prefix, filled by the caller][head of the record:
  // lay.ddata bytes][histogram][code table][counts] in ONE pinned allocation
  PinBuf pin;
  PinBuf dpin;                         // decompression: pinned copy of the decode table (source of its upload)
  PinBuf tagpin;                       // decompression: word a kernel clears when a device-resident record's sync tag is wrong
  bool tag_pending = false;            // ... to be looked at once the stream has been synchronised (lossless_tag_check)
  uint8_t *chead = nullptr;            // = pin.p + 8
  unsigned long long *pcounts = nullptr;  // [0..2] encoder state, [3] outlier count read back, [4] n_outliers to write
  PayloadLayout lay;
  uint64_t n_units = 0, n_outliers = 0;
  uint64_t outliers_needed = 0;  // set when lossless_compress returns kOutlierOverflow
  const uint64_t *d_oidx = nullptr;
  const int64_t *d_oval = nullptr;
  bool on_host = false;  // Huffman_Zstd: the whole record is in `host`
  // the encoder wrote the code units straight into the caller's record (lossless_compress:
  // `direct`): record_write() does not move them again
  bool units_in_place = false;
  std::vector<uint32_t> h_dtable;  // decompression: two-level decode table (source of an asynchronous upload)
  size_t record_size() const { return overflow ? ~(size_t)0 : (on_host ? host.size() : lay.total); }
};

namespace {

using ChunkFn = std::function<int(size_t, size_t, hipEvent_t)>;
int copy_any(void *dst, const void *src, size_t bytes, hipStream_t st, const ChunkFn *on_chunk = nullptr);  // (below)

// The single-pass encoder stages code table and symbols of a chunk in LDS.
inline bool lossless_sym16_ok(uint64_t dict, uint64_t chunk) {
  return dict <= 65536 && dict * 8 + chunk * 2 <= 140 * 1024;
}

// Host copy of the first bytes of the device-resident stream a decompression call is reading:
// header, record size and the leading part of the first record come out of ONE device-to-host
// copy instead of four synchronous ones (tens of microseconds each). Valid only inside
// mgh_decompress (HostPrefix guard).
struct HostPrefixState {
  const uint8_t *base = nullptr;
  std::vector<uint8_t> bytes;
};
inline HostPrefixState &host_prefix() {
  static thread_local HostPrefixState s;
  return s;
}
int aux_read(void *dst, const void *src, size_t bytes);  // (below, with the cache)
struct HostPrefix {
  HostPrefix(const void *dev, size_t size) {
    HostPrefixState &s = host_prefix();
    s.base = nullptr;
    const size_t want = std::min<size_t>(size, 320 * 1024);
    s.bytes.resize(want);
    // (through the cache's pinned buffer and its own stream once a cache exists: a hipMemcpy into
    // pageable memory is staged by the runtime, ~20 us more per call)
    if (aux_read(s.bytes.data(), dev, want) == MGH_SUCCESS)
      s.base = (const uint8_t *)dev;
  }
  ~HostPrefix() { host_prefix().base = nullptr; }
};
hipStream_t cache_copy_stream(int dev);                   // (below: the calling thread's copy stream on that device, or nullptr)
// device -> host copy that is served from the prefix where it can be
inline int dev_to_host(void *dst, const void *src, size_t bytes) {
  const HostPrefixState &s = host_prefix();
  const uint8_t *p = (const uint8_t *)src;
  if (s.base && p >= s.base && bytes <= s.bytes.size() && (size_t)(p - s.base) <= s.bytes.size() - bytes) {
    std::memcpy(dst, s.bytes.data() + (p - s.base), bytes);
    return MGH_SUCCESS;
  }
  return aux_read(dst, src, bytes);
}

// The small pieces of a device-resident record -- its head out of the context's pinned buffer (which
// the device reads directly), the outlier count, the two outlier lists, the synchronisation points
// -- in ONE launch instead of five asynchronous copies of ~7 us of host time each, which all sit
// between the encoder's result and the caller's return. Any byte alignment on either side.
struct RecordPieces {
  const uint8_t *src[5];
  uint8_t *dst[5];
  size_t bytes[5];
  int n;
  // decompression of a device-resident record: the 8 bytes that must be the tag of the
  // synchronisation-point section (any alignment); *tag_ok (pinned host memory) = 0 if they are not
  const uint8_t *tag_src;
  unsigned long long tag_want;
  unsigned *tag_ok;
};
__global__ void __launch_bounds__(256) k_record_pieces(RecordPieces P) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  if (P.tag_src && t == 0) {
    unsigned long long tag;
    __builtin_memcpy(&tag, P.tag_src, 8);
    if (tag != P.tag_want) *P.tag_ok = 0;
  }
  for (int i = 0; i < P.n; i++) {
    const uint8_t *sp = P.src[i];
    uint8_t *dp = P.dst[i];
    const size_t words = P.bytes[i] / 8;
    for (size_t w = t; w < words; w += stride) {
      unsigned long long v;
      __builtin_memcpy(&v, sp + 8 * w, 8);
      __builtin_memcpy(dp + 8 * w, &v, 8);
    }
    for (size_t b = words * 8 + t; b < P.bytes[i]; b += stride) dp[b] = sp[b];
  }
}

// Copy the record of the last lossless_compress() to dst (host or device memory, record_size()
// bytes). Asynchronous on st where the memory kinds allow it.
int record_write(mgh_lossless_ctx *c, void *dst, hipStream_t st, const uint64_t *size_prefix = nullptr) {
  // size_prefix: the 8 bytes in front of dst get *size_prefix -- for a device-resident record they
  // travel with the head of the record in ONE copy out of the pinned buffer
  char *d = (char *)dst;
  if (c->on_host) {
    if (size_prefix) {
      // (through the context's pinned buffer: the caller's variable may be gone before the queued copy runs)
      std::memcpy(c->chead - 8, size_prefix, 8);
      HL_HIP(hipMemcpyAsync(d - 8, c->chead - 8, 8, hipMemcpyDefault, st));
    }
    HL_HIP(hipMemcpyAsync(d, c->host.data(), c->host.size(), hipMemcpyDefault, st));
    return MGH_SUCCESS;
  }
  const PayloadLayout &L = c->lay;
  void *dev_head = nullptr;
  if (is_device_pointer_on(dst, c->dev) &&
      hipHostGetDevicePointer(&dev_head, c->chead - 8, 0) == hipSuccess && dev_head) {
    if (size_prefix) std::memcpy(c->chead - 8, size_prefix, 8);
    if (c->n_units && !c->units_in_place)
      HL_TRY(copy_any(d + L.ddata, c->units.p, c->n_units * 8, st));
    RecordPieces P{};
    auto piece = [&](const void *src, void *to, size_t bytes) {
      if (!bytes) return;
      P.src[P.n] = (const uint8_t *)src;
      P.dst[P.n] = (uint8_t *)to;
      P.bytes[P.n] = bytes;
      P.n++;
    };
    const uint8_t *dh = (const uint8_t *)dev_head;
    if (size_prefix) piece(dh, d - 8, 8 + L.ddata);
    else piece(dh + 8, d, L.ddata);
    // (the count out of the pinned buffer too: &c->pcounts[4] lies behind the head in the same allocation)
    piece(dh + ((const uint8_t *)&c->pcounts[4] - (c->chead - 8)), d + L.outlier_count, 8);
    piece(c->d_oidx, d + L.outlier_idx, c->n_outliers * 8);
    piece(c->d_oval, d + L.outliers, c->n_outliers * 8);
    if (c->use_sync) piece(c->sync.p, d + L.sync_tag, PayloadLayout::sync_bytes(c->n_chunks));
    size_t total = 0;
    for (int i = 0; i < P.n; i++) total += P.bytes[i];
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>(total / (256 * 64), 1), 512);
    k_record_pieces<<<blocks, 256, 0, st>>>(P);
    HL_HIP(hipGetLastError());
    return MGH_SUCCESS;
  }
  (void)hipGetLastError();
  if (size_prefix) {
    std::memcpy(c->chead - 8, size_prefix, 8);
    HL_HIP(hipMemcpyAsync(d - 8, c->chead - 8, 8 + L.ddata, hipMemcpyDefault, st));
  } else {
    HL_HIP(hipMemcpyAsync(d, c->chead, L.ddata, hipMemcpyDefault, st));
  }
  if (c->n_units && !c->units_in_place)
    HL_TRY(copy_any(d + L.ddata, c->units.p, c->n_units * 8, st));
  // (the count travels from pinned memory that outlives the asynchronous copy)
  HL_HIP(hipMemcpyAsync(d + L.outlier_count, &c->pcounts[4], 8, hipMemcpyDefault, st));
  if (c->n_outliers) {
    HL_HIP(hipMemcpyAsync(d + L.outlier_idx, c->d_oidx, c->n_outliers * 8, hipMemcpyDefault, st));
    HL_HIP(hipMemcpyAsync(d + L.outliers, c->d_oval, c->n_outliers * 8, hipMemcpyDefault, st));
  }
  if (c->use_sync)  // (tag + entries as the encoder left them: one piece)
    HL_TRY(copy_any(d + L.sync_tag, c->sync.p, PayloadLayout::sync_bytes(c->n_chunks), st));
  return MGH_SUCCESS;
}

// ocount: number of outliers, or (d_ocount != nullptr) read from the device together with the
// results of the encoder (one host synchronisation less) and checked against ocap.
// cap_units: upper bound for the code stream the caller is interested in (0: worst case).
// The stage runs in two halves so that the subdomain pipeline can queue the first half of
// subdomain k+1 (behind its decomposition, on its own stream) before it waits for subdomain k:
//   lossless_begin():  histogram kernel + its copy to the host, queued on st, no synchronisation;
//   lossless_finish(): waits for the histogram, builds the code on the host, encodes, reads the
//                      counts back (second synchronisation) and lays out the head of the record.
// lossless_compress() = both, for the stand-alone entry point.
struct LosslessJob {
  const int64_t *d_q = nullptr;
  uint64_t n = 0, dict = 0, chunk = 0;
  int lossless = MGH_LOSSLESS_HUFFMAN, zstd_level = 3;
  const uint64_t *d_oidx = nullptr;
  const int64_t *d_oval = nullptr;
  uint64_t ocount = 0;
  const uint64_t *d_ocount = nullptr;
  uint64_t ocap = ~(uint64_t)0, cap_units = 0;
  bool sym16 = false;
};

// Zeroing of the stage's device state (histogram bins, the encoder's chunk states, optionally a
// counter of the caller) in ONE launch. Queued by the subdomain pipeline in FRONT of the
// subdomain's decomposition, where the device is waiting for the host's launches anyway; three
// memsets between quantizer and encoder were ~15 us of every record's critical path.
__global__ void __launch_bounds__(256) k_zero_words(unsigned long long *a, size_t na, unsigned long long *b,
                                                    size_t nb, unsigned long long *c1) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (size_t i = t; i < na; i += stride) a[i] = 0;
  for (size_t i = t; i < nb; i += stride) b[i] = 0;
  if (c1 && t == 0) *c1 = 0;
}
int lossless_prepare(mgh_lossless_ctx *c, uint64_t n, uint64_t dict, uint64_t chunk, hipStream_t st,
                     unsigned long long *also_zero = nullptr) {
  c->prepared_n = 0;
  if (n == 0 || dict == 0 || dict > 16384 || chunk == 0 || chunk > (1u << 30)) return MGH_SUCCESS;  // (lossless_begin reports it)
  const size_t nchunk = (n - 1) / chunk + 1;
  HL_TRY(c->freq.ensure((dict * 4 + 7) / 8 * 8));
  HL_TRY(c->state.ensure((3 + nchunk) * 8));
  const size_t words = (dict * 4 + 7) / 8 + 3 + nchunk;
  k_zero_words<<<(unsigned)std::min<size_t>((words + 255) / 256, 256), 256, 0, st>>>(
      (unsigned long long *)c->freq.p, (dict * 4 + 7) / 8, (unsigned long long *)c->state.p, 3 + nchunk, also_zero);
  HL_HIP(hipGetLastError());
  c->prepared_n = n;
  c->prepared_dict = dict;
  c->prepared_chunk = chunk;
  return MGH_SUCCESS;
}

int lossless_begin(mgh_lossless_ctx *c, const LosslessJob &J, hipStream_t st) {
  const int64_t *d_q = J.d_q;
  const uint64_t n = J.n, dict = J.dict, chunk = J.chunk;
  const int lossless = J.lossless;
  const bool sym16 = J.sym16;
  // sym16: d_q points to uint16_t symbols (mgh_decompose_quantize_sym16) -- only with the
  // single-pass encoder (lossless_sym16_ok)
  if (lossless != MGH_LOSSLESS_HUFFMAN && lossless != MGH_LOSSLESS_HUFFMAN_ZSTD)
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: only Huffman and Huffman_Zstd are supported");
  if (n == 0 || dict == 0 || dict > 16384 || chunk == 0 || chunk > (1u << 30))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: n, dict_size (<= 16384) or chunk_size");
  if (lossless == MGH_LOSSLESS_HUFFMAN_ZSTD && !g_zstd.load())
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: libzstd.so.1 not found");
  const size_t nchunk = (n - 1) / chunk + 1;
  const bool prepared = c->prepared_n == n && c->prepared_dict == dict && c->prepared_chunk == chunk;
  c->prepared_n = 0;  // (good for one record)
  HL_TRY(c->freq.ensure((dict * 4 + 7) / 8 * 8));
  HL_TRY(c->code.ensure(dict * 8));
  HL_TRY(c->bits.ensure(nchunk * 8));
  HL_TRY(c->entry.ensure(nchunk * 8));
  HL_TRY(c->total.ensure(8));
  hl_debug("lossless_compress: begin");
  // the frequency counts are 32-bit (a subdomain of 2^32 symbols is 32 GB of int64: the domain
  // decomposer never produces one, but the stand-alone entry point could be handed one)
  if (n >= ((uint64_t)1 << 32))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless stage: more than 2^32 - 1 symbols in one record");
  if (!prepared) HL_HIP(hipMemsetAsync(c->freq.p, 0, dict * 4, st));
  const unsigned hblocks =
      (unsigned)std::min<size_t>((n + huff::kHistThreads - 1) / huff::kHistThreads, 2 * (size_t)hl_num_cu());
  if (sym16)
    huff::k_histogram<uint16_t><<<hblocks, huff::kHistThreads, dict * 4, st>>>((const uint16_t *)d_q, n, (int)dict,
                                                                               (unsigned *)c->freq.p);
  else
    huff::k_histogram<int64_t><<<hblocks, huff::kHistThreads, dict * 4, st>>>(d_q, n, (int)dict,
                                                                              (unsigned *)c->freq.p);
  HL_HIP(hipGetLastError());
  // (the encoder's chunk states are cleared here, behind the histogram kernel, so that nothing but
  // the code table stands between the code construction on the host and the encoder's launch)
  HL_TRY(c->state.ensure((3 + nchunk) * 8));
  if (!prepared) HL_HIP(hipMemsetAsync(c->state.p, 0, (3 + nchunk) * 8, st));
  hl_debug("lossless_compress: histogram kernel done");
  // pinned staging of this call (see PinBuf)
  PayloadLayout &L = c->lay;
  L.compute(nchunk, dict, 0, 0);  // (the offsets before the code units do not depend on the counts)
  const size_t o_head = 8, o_freq = (o_head + L.ddata + 15) / 16 * 16, o_code = o_freq + dict * 4,
               o_cnt = o_code + dict * 8;
  HL_TRY(c->pin.ensure(o_cnt + 64));
  uint8_t *const pinp = (uint8_t *)c->pin.p;
  c->chead = pinp + o_head;
  c->pcounts = (unsigned long long *)(pinp + o_cnt);
  unsigned *freq = (unsigned *)(pinp + o_freq);
  HL_HIP(hipMemcpyAsync(freq, c->freq.p, dict * 4, hipMemcpyDeviceToHost, st));
  return MGH_SUCCESS;
}

int lossless_finish(mgh_lossless_ctx *c, const LosslessJob &J, hipStream_t st, uint8_t *direct = nullptr,
                    size_t direct_cap = 0) {
  const int64_t *d_q = J.d_q;
  const uint64_t n = J.n, dict = J.dict, chunk = J.chunk, ocap = J.ocap, cap_units = J.cap_units;
  const int lossless = J.lossless, zstd_level = J.zstd_level;
  const uint64_t *d_oidx = J.d_oidx, *d_ocount = J.d_ocount;
  const int64_t *d_oval = J.d_oval;
  uint64_t ocount = J.ocount;
  const bool sym16 = J.sym16;
  // direct (memory of this device, 8-byte aligned, direct_cap bytes): where record_write() will be
  // asked to put this record. The single-pass encoder then writes the code units there itself --
  // the offset of the units inside a record does not depend on the counts -- instead of into a
  // buffer of the context from which record_write() copies them (512^3 f32: 150 MB moved twice).
  c->units_in_place = false;
  const size_t nchunk = (n - 1) / chunk + 1;
  PayloadLayout &L = c->lay;
  const size_t o_head = 8, o_freq = (o_head + L.ddata + 15) / 16 * 16, o_code = o_freq + dict * 4;
  uint8_t *const pinp = (uint8_t *)c->pin.p;
  unsigned *freq = (unsigned *)(pinp + o_freq);
  HL_HIP(hipStreamSynchronize(st));
  hl_debug("lossless_compress: histogram on the host");
  huff::Codebook &cb = c->codebook;
  try {
    huff::build_codebook(freq, (int)dict, cb);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, e.what());
  }
  hl_debug("lossless_compress: histogram + codebook done");
  // code table for the kernels: 32-bit entries when every code fits 27 bits (it practically
  // always does: half the LDS of the encoder, two workgroups per CU), else 64-bit entries
  // (codes longer than 27 bits -- the rarest symbols of a large subdomain -- become escape entries
  // into a list of 64-bit entries behind the table)
  size_t nlong = 0;
  if (cb.max_len > huff::kShortCodeBits)
    for (uint64_t k = 0; k < dict; k++) nlong += (cb.code[k] >> huff::kMaxCodeBits) > (uint64_t)huff::kShortCodeBits;
  const size_t o_long = (dict * 4 + 7) / 8 * 8;  // offset of the escape list behind the 32-bit table
  const bool short_codes = o_long + nlong * 8 <= dict * 8;  // (table + list fit the buffers sized for 64-bit entries)
  if (!short_codes) nlong = 0;
  if (short_codes) {
    uint32_t *c32 = (uint32_t *)(pinp + o_code);
    uint64_t *lg = (uint64_t *)(pinp + o_code + o_long);
    size_t j = 0;
    for (uint64_t k = 0; k < dict; k++) {
      const uint64_t len = cb.code[k] >> huff::kMaxCodeBits;
      if (len <= (uint64_t)huff::kShortCodeBits) {
        c32[k] = (uint32_t)(len << huff::kShortCodeBits) |
                 (uint32_t)(cb.code[k] & (((uint64_t)1 << huff::kShortCodeBits) - 1));
      } else {
        c32[k] = (huff::kEscapeLen << huff::kShortCodeBits) | (uint32_t)j;
        lg[j++] = cb.code[k];
      }
    }
    HL_HIP(hipMemcpyAsync(c->code.p, c32, o_long + nlong * 8, hipMemcpyHostToDevice, st));
  } else {
    std::memcpy(pinp + o_code, cb.code.data(), dict * 8);
    HL_HIP(hipMemcpyAsync(c->code.p, pinp + o_code, dict * 8, hipMemcpyHostToDevice, st));
  }
  // Synchronisation points for the decoder behind the record (PayloadLayout): with the single-pass
  // encoder, chunks of the size it keeps in registers, and streams of 4 bits per symbol or more
  // -- 256 bytes per chunk are 1.1 % of a chunk of 20 480 nine-bit codes, and the short codes of
  // a low-entropy stream re-synchronise within a symbol or two anyway. MGH_HUFF_SYNC=0: never.
  c->use_sync = false;
  c->n_chunks = nchunk;
  if (lossless == MGH_LOSSLESS_HUFFMAN && lossless_sym16_ok(dict, chunk) && chunk >= 1024 &&
      chunk <= (uint64_t)huff::kEncRun * huff::kEncThreads && env_get("MGH_HUFF_SYNC", 1) != 0) {
    c->use_sync = cb.total_bits >= 4 * n;
    if (c->use_sync) HL_TRY(c->sync.ensure(PayloadLayout::sync_bytes(nchunk)));
  }
  const size_t sync_bytes = c->use_sync ? PayloadLayout::sync_bytes(nchunk) : 0;
  // ---- serialize (Huffman.hpp:163-239): the small leading part on the host, the code units
  // and the outlier lists stay where they are until record_write() ----
  uint8_t *const out = c->chead;
  std::memset(out, 0, L.ddata);
  auto fetch_meta = [&]() -> int {  // bits_per_chunk[] and word_entry[] into the head
    HL_HIP(hipMemcpyAsync(out + L.huffmeta, c->bits.p, nchunk * 8, hipMemcpyDeviceToHost, st));
    HL_HIP(hipMemcpyAsync(out + L.huffmeta + nchunk * 8, c->entry.p, nchunk * 8,
                          hipMemcpyDeviceToHost, st));
    if (d_ocount) HL_HIP(hipMemcpyAsync(&c->pcounts[3], d_ocount, 8, hipMemcpyDeviceToHost, st));
    return MGH_SUCCESS;
  };
  unsigned long long units = 0;
  const size_t enc_lds = huff::encode_chain_lds(dict, short_codes ? 4 : 8, chunk);
  if (lossless_sym16_ok(dict, chunk) && chunk <= (1u << 24)) {
    // one pass: bit counts, unit offsets (decoupled look-back) and packing in the same kernel.
    // The stream is written into a buffer of cap_units; more than that means "not compressible".
    const unsigned long long worst = n + nchunk;  // a code is shorter than a unit
    unsigned long long cap = cap_units ? std::min<unsigned long long>(cap_units, worst) : worst;
    unsigned long long *units_dst = nullptr;
    // (the record may start at any byte: the encoder stores its units unaligned where it has to)
    if (direct && lossless == MGH_LOSSLESS_HUFFMAN && direct_cap > L.ddata + 64 + sync_bytes &&
        is_device_pointer_on(direct, c->dev)) {
      // (what does not fit behind the units -- outlier lists -- is the caller's capacity check)
      cap = std::min<unsigned long long>(cap, (direct_cap - L.ddata - 8 - sync_bytes) / 8);
      units_dst = (unsigned long long *)(direct + L.ddata);
      c->units_in_place = true;
    } else {
      HL_TRY(c->units.ensure(std::max<size_t>(cap, 1) * 8 + 8));
      units_dst = (unsigned long long *)c->units.p;
    }
    static std::atomic<uint64_t> once{0};
    if (hl_attr_pending(once)) {
      const int lim = 144 * 1024;
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_encode_chain<int64_t, uint64_t>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_encode_chain<uint16_t, uint64_t>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_encode_chain<int64_t, uint32_t>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_encode_chain<uint16_t, uint32_t>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      hl_attr_done(once);
    }
    auto enc = [&](auto sym_tag, auto code_tag) {
      using SYM = decltype(sym_tag);
      using CODE = decltype(code_tag);
      huff::k_encode_chain<SYM, CODE><<<(unsigned)nchunk, huff::kEncThreads, enc_lds, st>>>(
          (const SYM *)d_q, n, (int)chunk, (int)dict, nchunk, (const CODE *)c->code.p,
          (unsigned long long *)c->state.p, (unsigned long long *)c->bits.p,
          (unsigned long long *)c->entry.p, units_dst, cap, (int)nlong,
          c->use_sync ? (unsigned *)c->sync.p + 2 : nullptr, PayloadLayout::kSyncTag);
    };
    if (sym16 && short_codes) enc(uint16_t(), uint32_t());
    else if (sym16) enc(uint16_t(), uint64_t());
    else if (short_codes) enc(int64_t(), uint32_t());
    else enc(int64_t(), uint64_t());
    HL_HIP(hipGetLastError());
    unsigned long long *st3 = c->pcounts;
    HL_HIP(hipMemcpyAsync(st3, c->state.p, 24, hipMemcpyDeviceToHost, st));
    HL_TRY(fetch_meta());
    HL_HIP(hipStreamSynchronize(st));
    if ((st3[2] & 2) && c->units_in_place) {
      // a chunk needed the path that packs with atomics, which an unaligned destination cannot
      // take (k_encode_chain): once more into the context's own (aligned) buffer
      c->units_in_place = false;
      cap = cap_units ? std::min<unsigned long long>(cap_units, worst) : worst;
      HL_TRY(c->units.ensure(std::max<size_t>(cap, 1) * 8 + 8));
      units_dst = (unsigned long long *)c->units.p;
      HL_HIP(hipMemsetAsync(c->state.p, 0, (3 + nchunk) * 8, st));
      if (sym16 && short_codes) enc(uint16_t(), uint32_t());
      else if (sym16) enc(uint16_t(), uint64_t());
      else if (short_codes) enc(int64_t(), uint32_t());
      else enc(int64_t(), uint64_t());
      HL_HIP(hipGetLastError());
      HL_HIP(hipMemcpyAsync(st3, c->state.p, 24, hipMemcpyDeviceToHost, st));
      HL_TRY(fetch_meta());
      HL_HIP(hipStreamSynchronize(st));
    }
    if (d_ocount) ocount = c->pcounts[3];
    units = st3[1];
    c->overflow = (st3[2] & 1) != 0;
  } else {
    if (sym16) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: 16-bit symbols need the single-pass encoder");
    c->use_sync = false;
    if (short_codes)  // (these kernels read 64-bit entries)
      HL_HIP(hipMemcpyAsync(c->code.p, cb.code.data(), dict * 8, hipMemcpyHostToDevice, st));
    huff::k_chunk_bits<<<(unsigned)nchunk, 256, 0, st>>>(d_q, n, (int)chunk, (const uint64_t *)c->code.p,
                                                         (unsigned long long *)c->bits.p);
    huff::k_unit_offsets<<<1, 1024, 0, st>>>((const unsigned long long *)c->bits.p, nchunk,
                                             (unsigned long long *)c->entry.p,
                                             (unsigned long long *)c->total.p);
    HL_HIP(hipGetLastError());
    HL_HIP(hipMemcpyAsync(&c->pcounts[1], c->total.p, 8, hipMemcpyDeviceToHost, st));
    HL_TRY(fetch_meta());
    HL_HIP(hipStreamSynchronize(st));
    units = c->pcounts[1];
    if (d_ocount) ocount = c->pcounts[3];
    HL_TRY(c->units.ensure(std::max<size_t>(units, 1) * 8 + 8));
    HL_HIP(hipMemsetAsync(c->units.p, 0, units * 8, st));
    huff::k_encode<<<(unsigned)nchunk, 256, 0, st>>>(d_q, n, (int)chunk, (const uint64_t *)c->code.p,
                                                     (const unsigned long long *)c->entry.p,
                                                     (unsigned long long *)c->units.p);
    HL_HIP(hipGetLastError());
    c->overflow = false;
  }
  hl_debug("lossless_compress: encode launched");
  if (ocount > ocap) {
    // the caller re-runs the quantizer with buffers of this size (the reference re-allocates and
    // re-launches the same way: LinearQuantization.hpp:621-676)
    c->outliers_needed = ocount;
    return kOutlierOverflow;
  }
  c->on_host = false;
  if (c->overflow) return MGH_SUCCESS;  // record_size() says "larger than anything"
  L.compute(nchunk, dict, units, ocount, c->use_sync);
  auto put64 = [&](size_t off, uint64_t v) { std::memcpy(out + off, &v, 8); };
  auto put32 = [&](size_t off, int32_t v) { std::memcpy(out + off, &v, 4); };
  put64(L.primary_count, n);
  put32(8, (int32_t)dict);
  put32(12, (int32_t)chunk);
  put64(16, 2 * nchunk);
  put64(L.decodebook_size, 8 * (2 * 64) + 8 * dict);
  std::memcpy(out + L.decodebook, cb.first.data(), 8 * 64);
  std::memcpy(out + L.decodebook + 8 * 64, cb.entry.data(), 8 * 64);
  std::memcpy(out + L.decodebook + 8 * 128, cb.keys.data(), 8 * dict);
  put64(L.ddata_size, units);
  c->n_units = units;
  c->n_outliers = ocount;
  c->pcounts[4] = ocount;
  c->d_oidx = d_oidx;
  c->d_oval = d_oval;
  if (lossless == MGH_LOSSLESS_HUFFMAN_ZSTD) {
    // [size_t input_count][zstd frame] (Zstd.hpp:69-90): needs the whole record on the host
    std::vector<uint8_t> &full = c->host2;
    full.resize(L.total);
    HL_TRY(record_write(c, full.data(), st));
    HL_HIP(hipStreamSynchronize(st));
    const size_t bound = g_zstd.compressBound(full.size());
    c->host.resize(bound + 8);
    const size_t got = g_zstd.compress(c->host.data() + 8, bound, full.data(), full.size(), zstd_level);
    if (g_zstd.isError(got)) return hl_fail(MGH_ERR_DEVICE, "ZSTD_compress failed");
    const uint64_t in_size = full.size();
    std::memcpy(c->host.data(), &in_size, 8);
    c->host.resize(got + 8);
    c->on_host = true;
  }
  return MGH_SUCCESS;
}

int lossless_compress(mgh_lossless_ctx *c, const int64_t *d_q, uint64_t n, uint64_t dict,
                      uint64_t chunk, int lossless, int zstd_level, const uint64_t *d_oidx,
                      const int64_t *d_oval, uint64_t ocount, hipStream_t st,
                      const uint64_t *d_ocount = nullptr, uint64_t ocap = ~(uint64_t)0,
                      uint64_t cap_units = 0, bool sym16 = false, uint8_t *direct = nullptr,
                      size_t direct_cap = 0) {
  LosslessJob J;
  J.d_q = d_q; J.n = n; J.dict = dict; J.chunk = chunk; J.lossless = lossless; J.zstd_level = zstd_level;
  J.d_oidx = d_oidx; J.d_oval = d_oval; J.ocount = ocount; J.d_ocount = d_ocount; J.ocap = ocap;
  J.cap_units = cap_units; J.sym16 = sym16;
  HL_TRY(lossless_begin(c, J, st));
  return lossless_finish(c, J, st, direct, direct_cap);
}

// `payload` may be host or device memory: only the small leading part of the record is brought
// to the host, the code units and outlier lists go device-to-device (or host-to-device).
// sym16: decode to uint16_t symbols at d_q (the ring decoder only; *sym16 is cleared when another
// decoder had to be used and d_q holds int64 values).
// After a synchronisation of the stream lossless_decompress() ran on: was the tag of the
// synchronisation-point section of a device-resident record what it has to be?
int lossless_tag_check(mgh_lossless_ctx *c) {
  if (!c->tag_pending) return MGH_SUCCESS;
  c->tag_pending = false;
  if (*reinterpret_cast<volatile unsigned *>(c->tagpin.p) == 0)
    return hl_fail(MGH_ERR_FORMAT, "Huffman record: outlier lists");
  return MGH_SUCCESS;
}

int lossless_decompress(mgh_lossless_ctx *c, const uint8_t *payload, uint64_t size, int lossless,
                        int64_t *d_q, uint64_t n, uint64_t *ocount_out, hipStream_t st,
                        bool *sym16 = nullptr, bool sync_end = true) {
  const uint8_t *p = payload;
  uint64_t psize = size;
  bool on_dev = is_device_pointer(payload);
  if (lossless == MGH_LOSSLESS_HUFFMAN_ZSTD) {
    if (!g_zstd.load()) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: libzstd.so.1 not found");
    if (size < 8) return hl_fail(MGH_ERR_FORMAT, "zstd record truncated");
    const uint8_t *src = payload;
    if (on_dev) {
      c->host.resize(size);
      HL_HIP(hipMemcpy(c->host.data(), payload, size, hipMemcpyDeviceToHost));
      src = c->host.data();
    }
    uint64_t raw = 0;
    std::memcpy(&raw, src, 8);
    if (raw > ((uint64_t)1 << 40)) return hl_fail(MGH_ERR_FORMAT, "zstd record: implausible size");
    c->host2.resize(raw);
    const size_t got = g_zstd.decompress(c->host2.data(), raw, src + 8, size - 8);
    if (g_zstd.isError(got) || got != raw) return hl_fail(MGH_ERR_FORMAT, "ZSTD_decompress failed");
    p = c->host2.data();
    psize = raw;
    on_dev = false;
  } else if (lossless != MGH_LOSSLESS_HUFFMAN) {
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: only Huffman and Huffman_Zstd are supported");
  }
  // host copy of bytes [off, off + bytes) of the record
  std::vector<uint8_t> &head = c->head;
  auto need = [&](size_t off, size_t bytes) { return off <= psize && bytes <= psize - off; };
  auto fetch = [&](size_t upto) -> int {  // make head cover [0, upto)
    if (head.size() >= upto) return MGH_SUCCESS;
    const size_t have = head.size();
    head.resize(upto);
    if (on_dev) HL_TRY(dev_to_host(head.data() + have, p + have, upto - have));
    else std::memcpy(head.data() + have, p + have, upto - have);
    return MGH_SUCCESS;
  };
  head.clear();
  if (!need(0, 24)) return hl_fail(MGH_ERR_FORMAT, "Huffman record truncated");
  // (a record in device memory: one copy that covers the whole leading part for the default
  // parameters instead of one per field -- every synchronous copy costs tens of microseconds)
  if (on_dev) HL_TRY(fetch(std::min<size_t>(psize, 24 + 16 * ((n - 1) / 20480 + 1) + 16 + 8 * 128 + 8 * 8192 + 16)));
  else HL_TRY(fetch(24));
  uint64_t primary = 0, huffmeta_size = 0;
  int32_t dict = 0, chunk = 0;
  std::memcpy(&primary, head.data(), 8);
  std::memcpy(&dict, head.data() + 8, 4);
  std::memcpy(&chunk, head.data() + 12, 4);
  std::memcpy(&huffmeta_size, head.data() + 16, 8);
  if (primary != n || dict <= 0 || dict > 16384 || chunk <= 0 ||
      huffmeta_size != 2 * ((n - 1) / (uint64_t)chunk + 1))
    return hl_fail(MGH_ERR_FORMAT, "Huffman record: header does not match the subdomain");
  const size_t nchunk = huffmeta_size / 2;
  PayloadLayout L;
  L.compute(nchunk, (size_t)dict, 0, 0);
  if (!need(0, L.ddata)) return hl_fail(MGH_ERR_FORMAT, "Huffman record truncated");
  HL_TRY(fetch(L.ddata));
  uint64_t dbsize = 0, units = 0;
  std::memcpy(&dbsize, head.data() + L.decodebook_size, 8);
  if (dbsize != 8 * 128 + 8 * (uint64_t)dict) return hl_fail(MGH_ERR_FORMAT, "Huffman record: decodebook size");
  std::memcpy(&units, head.data() + L.ddata_size, 8);
  if (units > (psize - L.ddata) / 8) return hl_fail(MGH_ERR_FORMAT, "Huffman record truncated");
  const size_t o_oc = L.ddata + 8 * units;
  if (!need(o_oc, 8)) return hl_fail(MGH_ERR_FORMAT, "Huffman record truncated");
  uint64_t ocount = 0;
  // Behind the outlier lists: nothing, or the synchronisation points of the decoder (PayloadLayout;
  // 8 mod 16 bytes where the lists are 0 mod 16).
  const size_t sync_bytes = PayloadLayout::sync_bytes(nchunk);
  const size_t rem = psize - o_oc - 8;
  bool has_sync = rem % 16 == 8 && rem >= sync_bytes;
  if (on_dev) {
    // the record ends with the two outlier arrays (and that section): their length follows from
    // the record size (saves a synchronous 8-byte copy from the device)
    if (rem % 16 != 0 && !has_sync) return hl_fail(MGH_ERR_FORMAT, "Huffman record: outlier lists");
    ocount = (rem - (has_sync ? sync_bytes : 0)) / 16;
  } else {
    std::memcpy(&ocount, p + o_oc, 8);
    if (has_sync && (ocount > (rem - sync_bytes) / 16 || rem - 16 * ocount != sync_bytes)) has_sync = false;
  }
  if (ocount > rem / 16) return hl_fail(MGH_ERR_FORMAT, "Huffman record truncated");
  const size_t o_oidx = o_oc + 8, o_oval = o_oidx + 8 * ocount;
  const size_t o_sync = o_oval + 8 * ocount + 8;  // (behind the tag)
  if (has_sync && !on_dev) {
    uint64_t tag = 0;
    std::memcpy(&tag, p + o_sync - 8, 8);
    if (tag != PayloadLayout::kSyncTag) has_sync = false;
  }
  if (env_get("MGH_HUFF_SYNC_DECODE", 1) == 0) has_sync = false;  // cross-check: decode without them
  // the chunk entries must stay inside the unit array (they index it in the decoder)
  {
    const uint64_t *bits = reinterpret_cast<const uint64_t *>(head.data() + L.huffmeta);
    const uint64_t *ent = bits + nchunk;
    for (size_t k = 0; k < nchunk; k++)
      if (ent[k] > units || (bits[k] + 63) / 64 > units - ent[k])
        return hl_fail(MGH_ERR_FORMAT, "Huffman record: chunk outside the code stream");
  }
  hl_debug("lossless_decompress: record head parsed");
  HL_TRY(c->bits.ensure(nchunk * 8));
  HL_TRY(c->entry.ensure(nchunk * 8));
  HL_TRY(c->tables.ensure(dbsize));
  HL_TRY(c->oidx.ensure(std::max<size_t>(ocount, 1) * 8));
  HL_TRY(c->oval.ensure(std::max<size_t>(ocount, 1) * 8));
  // (a record in device memory: these go device-to-device from the record itself, not back up
  // from the pageable host copy)
  const uint8_t *meta_src = on_dev ? p : head.data();
  // (a record on this device: chunk table, decodebook and outlier lists in ONE launch -- five
  // queued copies of ~5 us each stood in front of every decoder launch)
  const bool pieces = on_dev && is_device_pointer_on(p, c->dev);
  if (pieces) {
    RecordPieces P{};
    auto piece = [&](const void *src, void *to, size_t bytes) {
      if (!bytes) return;
      P.src[P.n] = (const uint8_t *)src;
      P.dst[P.n] = (uint8_t *)to;
      P.bytes[P.n] = bytes;
      P.n++;
    };
    piece(p + L.huffmeta, c->bits.p, nchunk * 8);
    piece(p + L.huffmeta + nchunk * 8, c->entry.p, nchunk * 8);
    piece(p + L.decodebook, c->tables.p, dbsize);
    piece(p + o_oidx, c->oidx.p, ocount * 8);
    piece(p + o_oval, c->oval.p, ocount * 8);
    if (has_sync) {
      // The section was recognised by the record's size alone (no host copy of its tag): the kernel
      // looks at the tag and clears a pinned word if it is not one; whoever synchronises the stream
      // next turns that into MGH_ERR_FORMAT (lossless_tag_check) -- a damaged record must not decode
      // quietly with arbitrary synchronisation points.
      HL_TRY(c->tagpin.ensure(64));
      void *dev_word = nullptr;
      if (hipHostGetDevicePointer(&dev_word, c->tagpin.p, 0) == hipSuccess && dev_word) {
        *reinterpret_cast<volatile unsigned *>(c->tagpin.p) = 1;
        P.tag_src = p + o_sync - 8;
        P.tag_want = PayloadLayout::kSyncTag;
        P.tag_ok = (unsigned *)dev_word;
        c->tag_pending = true;
      } else {
        (void)hipGetLastError();
      }
    }
    size_t tot = 0;
    for (int i = 0; i < P.n; i++) tot += P.bytes[i];
    k_record_pieces<<<(unsigned)std::min<size_t>(std::max<size_t>(tot / (256 * 64), 1), 512), 256, 0, st>>>(P);
    HL_HIP(hipGetLastError());
  } else {
    HL_HIP(hipMemcpyAsync(c->bits.p, meta_src + L.huffmeta, nchunk * 8, hipMemcpyDefault, st));
    HL_HIP(hipMemcpyAsync(c->entry.p, meta_src + L.huffmeta + nchunk * 8, nchunk * 8, hipMemcpyDefault, st));
    HL_HIP(hipMemcpyAsync(c->tables.p, meta_src + L.decodebook, dbsize, hipMemcpyDefault, st));
  }
  static const bool serial_decode = env_get("MGH_HUFF_SERIAL_DECODE", 0) != 0;  // cross-check
  static const bool par_decode = env_get("MGH_HUFF_PAR_DECODE", 0) != 0;           // cross-check
  int book_max_len = 0;  // longest code of the decodebook (unused lengths carry first = 2^64-1)
  {
    const uint64_t *first = reinterpret_cast<const uint64_t *>(head.data() + L.decodebook);
    for (int l = 1; l < 64; l++)
      if (first[l] != ~(uint64_t)0) book_max_len = l;
  }
  // (the ring decoder keeps more than 32 bits in its bit buffer: codes of up to 32 bits)
  const bool ring_decode = !serial_decode && !par_decode && (size_t)chunk >= 1024 && (size_t)chunk <= (1u << 24) &&
                           dict <= 65536 && book_max_len <= 32;
  // A device-resident record is decoded where it is (512^3: 150 MB not copied) -- by the ring
  // decoder wherever its code units start (load_unit), by the others when they start 8-byte
  // aligned. The decoders peek one unit past the stream; in the record that is the outlier count
  // -- the peeked bits lie beyond the last code of the last chunk and never reach a symbol (every
  // chunk stops at its bit count).
  // (only memory of the device the decoder runs on: a record on another GPU is copied over)
  const bool units_in_place = on_dev && units && (ring_decode || ((uintptr_t)(p + L.ddata) & 7) == 0) &&
                              is_device_pointer_on(p, c->dev);
  const unsigned long long *d_units = (const unsigned long long *)c->units.p;
  bool units_follow = false;
  if (units_in_place) {
    d_units = (const unsigned long long *)(p + L.ddata);
  } else {
    HL_TRY(c->units.ensure((units + 1) * 8));  // (not for units decoded in place: 8 N bytes a lane would hold for nothing)
    d_units = (const unsigned long long *)c->units.p;
    // A large record in HOST memory is decoded while it arrives (below: the ring decoder's launches
    // follow the pieces of the copy, which runs on the cache's copy stream); everything else is
    // copied here, in stream order. (A record in pageable memory travels through the pinned ring.)
    units_follow = !on_dev && ring_decode && units * 8 >= ((size_t)32 << 20) && cache_copy_stream(c->dev) &&
                   env_get("MGH_HL_DECODE_FOLLOWS", 1) != 0;
    if (units && !units_follow) HL_TRY(copy_any(c->units.p, p + L.ddata, units * 8, st));
    HL_HIP(hipMemsetAsync((char *)c->units.p + units * 8, 0, 8, st));  // (the decoder peeks one unit ahead)
  }
  if (ocount && !pieces) {
    HL_HIP(hipMemcpyAsync(c->oidx.p, p + o_oidx, ocount * 8, hipMemcpyDefault, st));
    HL_HIP(hipMemcpyAsync(c->oval.p, p + o_oval, ocount * 8, hipMemcpyDefault, st));
  }
  hl_debug("lossless_decompress: uploads done");
  const unsigned long long *tab = (const unsigned long long *)c->tables.p;
  // prefix table as large as LDS allows next to the 16-bit keys (15 bits for dict = 8192)
  int tb = 15;
  const size_t lds_keys_ring = ((size_t)dict * 2 + 7) / 8 * 8 + 16 * 64 * 8;
  while (tb > 8 && ((size_t)4 << tb) + lds_keys_ring > 154 * 1024) tb--;
  tb = std::max(8, std::min(tb, (int)env_get("MGH_HUFF_TB", tb)));  // developer switch
  const size_t lds = ((size_t)4 << tb) + lds_keys_ring;
  static std::atomic<uint64_t> once{0};
  if (hl_attr_pending(once)) {
    HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    hl_attr_done(once);
  }
  hl_debug("lossless_decompress: uploads done (units, tables, outliers)");
  if (ring_decode) {
    // parallel decoding inside the chunks: two-level table from the decodebook (host, microseconds),
    // code units through per-lane LDS rings. 16 waves per workgroup when the table leaves room.
    const uint64_t *book = reinterpret_cast<const uint64_t *>(head.data() + L.decodebook);
    int rtb = 12;
    rtb = std::max(8, std::min(14, (int)env_get("MGH_HUFF_TB", rtb)));  // developer switch
    const size_t lds_cap = 150 * 1024;
    const size_t per_wave = huff::decode_ring_lds(0, 1);
    // (kept in the context: the upload below is asynchronous and must not outlive its source)
    std::vector<uint32_t> &dt = c->h_dtable;
    // Records with synchronisation points and SHORT codes: the decoder that takes two codes per
    // root-table slot where both fit its 12 bits (k_decode_sync). 512^3 f32, int64 output, same box:
    // 5.7 bits per symbol 0.79 against 0.85 ms with k_decode_ring's single-symbol steps; 7.4 bits 0.98
    // against 0.87, 9.1 bits (the benchmark's field at 1e-3) 0.97 against 0.79 -- pairs no longer fit
    // and the wider entries only cost. MGH_HUFF_PAIR: 0 never, 1 up to 6.5 bits per symbol (default),
    // 2 whenever the record has the points (cross-check).
    const long pair_env = env_get("MGH_HUFF_PAIR", 1);
    const bool pair_decode = has_sync && (size_t)chunk <= 65535 &&
                             (pair_env == 2 || (pair_env == 1 && (double)units * 64.0 <= 6.5 * (double)n));
    dt = huff::build_decode_table(book, book + 64, book + 128, (int)dict, rtb,
                                  (lds_cap - 8 * per_wave) / 4 - (pair_decode ? ((size_t)1 << rtb) : 0));
    if (pair_decode) dt = huff::make_pair_table(dt, rtb);
    const int waves = huff::decode_ring_lds(dt.size(), 16) <= lds_cap ? 16 : 8;
    HL_TRY(c->dtable.ensure(dt.size() * 4));
    // (out of pinned memory: a copy from pageable memory is staged synchronously, ~15 us)
    HL_TRY(c->dpin.ensure(dt.size() * 4));
    std::memcpy(c->dpin.p, dt.data(), dt.size() * 4);
    HL_HIP(hipMemcpyAsync(c->dtable.p, c->dpin.p, dt.size() * 4, hipMemcpyHostToDevice, st));
    static std::atomic<uint64_t> once3{0};
    if (hl_attr_pending(once3)) {
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_ring<int64_t>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_ring<uint16_t>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      hl_attr_done(once3);
    }
    // synchronisation points of the encoder, if the record has them: read where they lie in a
    // record on this device (any alignment), else from a copy
    const unsigned *d_sync = nullptr;
    if (has_sync && (size_t)chunk <= 65535) {
      if (on_dev && is_device_pointer_on(p, c->dev)) {
        d_sync = (const unsigned *)(p + o_sync);
      } else {
        if (on_dev) {  // (a record on another device: its tag has not been looked at yet)
          uint64_t tag = 0;
          HL_TRY(aux_read(&tag, p + o_sync - 8, 8));
          if (tag != PayloadLayout::kSyncTag) return hl_fail(MGH_ERR_FORMAT, "Huffman record: outlier lists");
        }
        HL_TRY(c->sync.ensure(sync_bytes - 8));
        HL_HIP(hipMemcpyAsync(c->sync.p, p + o_sync, sync_bytes - 8, hipMemcpyDefault, st));
        d_sync = (const unsigned *)c->sync.p;
      }
    }
    if (pair_decode && !d_sync) return hl_fail(MGH_ERR_DEVICE, "lossless_decompress: pair table without synchronisation points");
    if (pair_decode) {
      static std::atomic<uint64_t> once4{0};
      if (hl_attr_pending(once4)) {
        HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_sync<int64_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_sync<uint16_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        hl_attr_done(once4);
      }
    }
    const size_t lds_b = huff::decode_ring_lds(dt.size(), waves);
    const bool out16 = sym16 && *sym16;
    // MGH_HUFF_LEAN=1: the writing pass of records with synchronisation points without divergent
    // control flow (k_decode_lean: half the instructions per step -- and the same 0.51 ms at 512^3 as
    // k_decode_ring; it carries the ablation switches MGH_HUFF_DBG that say where the time goes,
    // profiles/NOTES.md round 6). Off by default: no gain, one kernel less in the default path.
    const bool lean_decode = d_sync && !pair_decode && env_get("MGH_HUFF_LEAN", 0) != 0;
    if (lean_decode) {
      static std::atomic<uint64_t> once5{0};
      if (hl_attr_pending(once5)) {
        HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_lean<int64_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_lean<uint16_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        hl_attr_done(once5);
      }
    }
    // chunks [c0, c1): the kernels index everything by chunk, so a range is the same launch with
    // the per-chunk arrays, the output and the symbol count moved up by c0 chunks
    auto launch_range = [&](size_t c0, size_t c1) -> int {
      if (c1 <= c0) return MGH_SUCCESS;
      const size_t cnt = c1 - c0, n_r = n - c0 * (size_t)chunk;
      const unsigned grid = (unsigned)((cnt + waves - 1) / waves);
      const unsigned long long *bits_r = (const unsigned long long *)c->bits.p + c0;
      const unsigned long long *ent_r = (const unsigned long long *)c->entry.p + c0;
      const unsigned *sync_r = d_sync ? reinterpret_cast<const unsigned *>(reinterpret_cast<const uint8_t *>(d_sync) + c0 * huff::kSyncLanes * 4) : nullptr;
      int64_t *q64 = d_q + c0 * (size_t)chunk;
      uint16_t *q16 = (uint16_t *)d_q + c0 * (size_t)chunk;
      if (pair_decode) {
        if (out16)
          huff::k_decode_sync<uint16_t><<<grid, 64 * waves, lds_b, st>>>(d_units, bits_r, ent_r, cnt, chunk, n_r, dict, rtb,
                                                                         (const unsigned *)c->dtable.p, (unsigned)dt.size(),
                                                                         tab, tab + 64, tab + 128, q16, sync_r);
        else
          huff::k_decode_sync<int64_t><<<grid, 64 * waves, lds_b, st>>>(d_units, bits_r, ent_r, cnt, chunk, n_r, dict, rtb,
                                                                        (const unsigned *)c->dtable.p, (unsigned)dt.size(),
                                                                        tab, tab + 64, tab + 128, q64, sync_r);
      } else if (lean_decode && out16) {
        huff::k_decode_lean<uint16_t><<<grid, 64 * waves, lds_b, st>>>(d_units, bits_r, ent_r, cnt, chunk, n_r, dict, rtb | ((int)env_get("MGH_HUFF_DBG", 0) << 8),
                                                                       (const unsigned *)c->dtable.p, (unsigned)dt.size(),
                                                                       tab, tab + 64, tab + 128, q16, sync_r);
      } else if (lean_decode) {
        huff::k_decode_lean<int64_t><<<grid, 64 * waves, lds_b, st>>>(d_units, bits_r, ent_r, cnt, chunk, n_r, dict, rtb | ((int)env_get("MGH_HUFF_DBG", 0) << 8),
                                                                      (const unsigned *)c->dtable.p, (unsigned)dt.size(),
                                                                      tab, tab + 64, tab + 128, q64, sync_r);
      } else if (out16) {
        huff::k_decode_ring<uint16_t><<<grid, 64 * waves, lds_b, st>>>(d_units, bits_r, ent_r, cnt, chunk, n_r, dict, rtb,
                                                                       (const unsigned *)c->dtable.p, (unsigned)dt.size(),
                                                                       tab, tab + 64, tab + 128, q16, sync_r);
      } else {
        huff::k_decode_ring<int64_t><<<grid, 64 * waves, lds_b, st>>>(d_units, bits_r, ent_r, cnt, chunk, n_r, dict, rtb,
                                                                      (const unsigned *)c->dtable.p, (unsigned)dt.size(),
                                                                      tab, tab + 64, tab + 128, q64, sync_r);
      }
      HL_HIP(hipGetLastError());
      return MGH_SUCCESS;
    };
    if (units_follow) {
      // the record's code units on the copy stream, piece by piece; behind every piece the chunks
      // whose units (and the one unit the decoder peeks at behind them) have landed are decoded on st
      const uint64_t *h_bits = reinterpret_cast<const uint64_t *>(head.data() + L.huffmeta);
      const uint64_t *h_ent = h_bits + nchunk;
      size_t c_done = 0;
      const size_t total_b = units * 8;
      const ChunkFn on_piece = [&](size_t off, size_t nb, hipEvent_t landed) -> int {
        const uint64_t have = (off + nb) / 8;  // units of the record on the device
        size_t c_hi = c_done;
        if (off + nb >= total_b) {
          c_hi = nchunk;
        } else {
          while (c_hi < nchunk && h_ent[c_hi] + (h_bits[c_hi] + 63) / 64 + 1 <= have) c_hi++;
        }
        if (c_hi > c_done) {
          HL_HIP(hipStreamWaitEvent(st, landed, 0));
          HL_TRY(launch_range(c_done, c_hi));
          c_done = c_hi;
        }
        return MGH_SUCCESS;
      };
      // (the copy stream must not run ahead of what st has queued in front: the small uploads above
      // are independent of the units; the units buffer itself is free -- the caller drained st)
      HL_TRY(copy_any(c->units.p, p + L.ddata, total_b, cache_copy_stream(c->dev), &on_piece));
      if (c_done < nchunk) return hl_fail(MGH_ERR_DEVICE, "lossless_decompress: chunks left behind the last piece");
    } else {
      HL_TRY(launch_range(0, nchunk));
    }
    HL_HIP(hipGetLastError());
  } else if (!serial_decode && (size_t)chunk >= 1024) {
    if (sym16) *sym16 = false;
    // parallel decoding inside the chunks (one wave per chunk)
    static std::atomic<uint64_t> once2{0};
    if (hl_attr_pending(once2)) {
      HL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(huff::k_decode_par),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      hl_attr_done(once2);
    }
    // 14-bit prefix table at most, so that the write-out staging (4 KiB per wave) fits beside it
    while (tb > 8 && ((size_t)4 << tb) + ((size_t)dict + 3) / 4 * 8 + huff::kParWaves * 64 * huff::kParBatch * 2 >
                         150 * 1024)
      tb--;
    const size_t lds_par = ((size_t)4 << tb) + ((size_t)dict + 3) / 4 * 8 + huff::kParWaves * 64 * huff::kParBatch * 2;
    huff::k_decode_par<<<(unsigned)((nchunk + huff::kParWaves - 1) / huff::kParWaves),
                         64 * huff::kParWaves, lds_par, st>>>(
        d_units, (const unsigned long long *)c->bits.p,
        (const unsigned long long *)c->entry.p, nchunk, chunk, n, dict, tb, tab, tab + 64, tab + 128, d_q);
  } else {
    if (sym16) *sym16 = false;
    huff::k_decode<<<(unsigned)((nchunk + 63) / 64), 64, lds, st>>>(
        d_units, (const unsigned long long *)c->bits.p,
        (const unsigned long long *)c->entry.p, nchunk, chunk, n, dict, tb, tab, tab + 64, tab + 128, d_q);
  }
  HL_HIP(hipGetLastError());
  hl_debug("lossless_decompress: decode launched");
  // the host payload may go away when we return (the subdomain pipeline keeps it, and the context's
  // host-side sources, alive until the lane has drained: sync_end = false)
  *ocount_out = ocount;
  if (sync_end) {
    HL_HIP(hipStreamSynchronize(st));
    return lossless_tag_check(c);
  }
  return MGH_SUCCESS;
}

// ---- domain decomposition (host logic) ------------------------------------------------------
struct Decomposer {
  int D = 0;
  std::vector<uint64_t> shape;
  bool decomposed = false;
  int method = MGH_DD_MAXDIM;
  uint64_t dim = 0, size = 0;        // MaxDim: (dim, size); Block: size; Variable: dim
  std::vector<uint64_t> var_sizes;   // Variable
  uint64_t num = 1;

  std::vector<uint64_t> dim_num_subdomain() const {  // DomainDecomposer.hpp:90-103
    std::vector<uint64_t> r(D, 1);
    if (method == MGH_DD_MAXDIM || method == MGH_DD_VARIABLE) r[dim] = num;
    else for (int d = 0; d < D; d++) r[d] = (shape[d] - 1) / size + 1;
    return r;
  }
  std::vector<uint64_t> dim_subdomain_id(uint64_t id) const {  // :105-113
    const auto nd = dim_num_subdomain();
    std::vector<uint64_t> r(D);
    for (int d = D - 1; d >= 0; d--) {
      r[d] = id % nd[d];
      id /= nd[d];
    }
    return r;
  }
  std::vector<uint64_t> subdomain_shape(uint64_t id) const {  // :124-168
    if (!decomposed) return shape;
    std::vector<uint64_t> r = shape;
    if (method == MGH_DD_MAXDIM) {
      r[dim] = id < shape[dim] / size ? size : shape[dim] % size;
    } else if (method == MGH_DD_BLOCK) {
      const auto sid = dim_subdomain_id(id);
      for (int d = 0; d < D; d++) r[d] = sid[d] < shape[d] / size ? size : shape[d] % size;
    } else {
      r[dim] = var_sizes[id];
    }
    return r;
  }
  std::vector<uint64_t> subdomain_offset(uint64_t id) const {  // :115-122, 690-700
    std::vector<uint64_t> r(D, 0);
    if (!decomposed) return r;
    if (method == MGH_DD_MAXDIM) {
      r[dim] = id * size;
    } else if (method == MGH_DD_BLOCK) {
      const auto sid = dim_subdomain_id(id);
      for (int d = 0; d < D; d++) r[d] = sid[d] * size;
    } else {
      for (uint64_t k = 0; k < id; k++) r[dim] += var_sizes[k];
    }
    return r;
  }
  // A subdomain is one contiguous run of the full array when it spans every dimension but the
  // slowest completely; then it can be used in place (no copy) if the array is device memory.
  bool contiguous(uint64_t id) const {
    const auto ext = subdomain_shape(id);
    for (int d = 1; d < D; d++)
      if (ext[d] != shape[d]) return false;
    return true;
  }
  uint64_t linear_offset(uint64_t id) const {
    uint64_t inner = 1;
    for (int d = 1; d < D; d++) inner *= shape[d];
    return subdomain_offset(id)[0] * inner;
  }
  bool all_contiguous() const {
    for (uint64_t id = 0; id < num; id++)
      if (!contiguous(id)) return false;
    return true;
  }
  uint64_t max_subdomain_elems() const {
    uint64_t m = 0;
    for (uint64_t id = 0; id < num; id++) {
      uint64_t c = 1;
      for (uint64_t e : subdomain_shape(id)) c *= e;
      m = std::max(m, c);
    }
    return m;
  }
};

// Device bytes the REFERENCE plans for one subdomain of this shape -- its formula, so that the
// MaxDim / Block auto-splits land on the reference's subdomain sizes
// (DomainDecomposer::EstimateMemoryFootprint, DomainDecomposer.hpp:24-69, with
// Hierarchy.hpp:420-, Compressor.hpp:84-116, DataRefactor.hpp:50-70, LinearQuantization.hpp:547-552,
// Lossless.hpp:58-71, HuffmanWorkspace.hpp:58-92):
//   [input N T + output N 8 + ratio 8 + hierarchy tables]  (x 2 with prefetch)
//   + refactoring workspace prod(n_d + 2) T (twice for D > 3) + quantizers + Huffman workspace
//     (outlier lists 16 N ratio, codes 8 N, chunk tables 24 nchunk, code-book scratch ~ 80 dict)
//     + the int64 array 8 N when T is narrower than 8 bytes.
// Left out, because they depend on the reference's runtime and are a few KB: the allocation pitch
// of the fastest dimension (hipMallocPitch), the radix sort's temporary storage for `dict`
// keys, 2 (warps-per-block x CUs + 1) status words. This implementation itself needs less
// (no (n+2)^D workspace, 16-bit symbols), so following the reference only means splitting earlier.
size_t estimate_footprint(const std::vector<uint64_t> &shape, size_t elem, const mgh_config &cfg,
                          bool prefetch) {
  const int D = (int)shape.size();
  double n = 1, ws = 1;
  for (uint64_t e : shape) {
    n *= (double)e;
    ws *= (double)(e + 2);
  }
  const double ratio = cfg.estimate_outlier_ratio;
  // levels: every dim is halved until the smallest reaches 2 (Hierarchy.hpp:428-446)
  int L = 64;
  for (uint64_t e : shape) {
    int k = 0;
    for (uint64_t m = e; m > 2; m = m / 2 + 1) k++;
    L = std::min(L, k);
  }
  double hier = 0;  // per level and dim: shape words, ranges, coordinates, dist, ratio, am, bm, volumes
  for (int l = 0; l <= L; l++) {
    for (uint64_t e : shape) {
      uint64_t m = e;
      for (int k = 0; k < L - l; k++) m = m / 2 + 1;
      hier += 6.0 * (double)(m + 1) * elem;
    }
    hier += (double)D * 8 * 2;
  }
  double b = n * elem + n * 8 + ratio * 8 + hier;
  if (prefetch) b *= 2;
  const double dict = (double)cfg.huff_dict_size;
  const double nchunk = std::floor((n - 1) / (double)cfg.huff_block_size) + 1;
  double lossless = 8 + n * ratio * 16 + dict * 4 + dict * 8 + (8 * 128 + 8 * dict) + n * 8 + 3 * nchunk * 8 +
                    4 + dict * 4 + dict * 8 + 4 * dict * 4 + 6 * dict * 4 + 8 * dict + 64;
  double comp = ws * elem * (D > 3 ? 2 : 1) + elem + (L + 1) * (double)elem + lossless + elem;
  if (8 > elem) comp += 8 * n;
  return (size_t)(b + comp);
}

// Device bytes THIS implementation keeps for a compression with `nlanes` pipeline lanes and
// `nbufs` input buffers (upper bound): per lane the quantized array (8 N: int64 where the 16-bit
// symbols do not apply), its level-linearised copy, the outlier lists (16 per estimated outlier),
// the hierarchy's workspace (levels below the top, per-slice vectors; the generic N-D path keeps
// three whole arrays) and the lossless stage's code units (a subdomain that does not compress
// below its own size is stored raw); per input buffer one dense subdomain.
size_t own_resident_bytes(int D, uint64_t max_elems, size_t elem, const mgh_config &cfg, uint64_t ocap,
                          int nlanes, int nbufs) {
  const double n = (double)max_elems;
  const double hier = (D <= 3 ? 0.5 : D == 4 ? 1.5 : 4.0) * n * (double)elem;
  const double lane = 8 * n + (cfg.reorder ? 8 * n : 0) + 16.0 * (double)ocap + hier + (n * (double)elem + 4096) +
                      (double)(64 << 20);  // (tables, chunk states, synchronisation points, allocator granularity)
  return (size_t)(lane * nlanes + (double)nbufs * n * (double)elem);
}

int make_decomposer(Decomposer &dd, int D, const uint64_t *shape, size_t elem, const mgh_config &cfg) {
  dd.D = D;
  dd.shape.assign(shape, shape + D);
  size_t free_b = 0, total_b = 0;
  HL_HIP(hipMemGetInfo(&free_b, &total_b));
  const size_t avail = std::min<size_t>(free_b, cfg.max_memory_footprint);
  auto need = [&](const std::vector<uint64_t> &s, bool prefetch) {
    return estimate_footprint(s, elem, cfg, prefetch) >= avail;
  };
  dd.method = cfg.domain_decomposition;
  if (!need(dd.shape, false) && dd.method != MGH_DD_BLOCK && dd.method != MGH_DD_VARIABLE) {
    dd.decomposed = false;  // DomainDecomposer.hpp:303-311
    dd.dim = 0;
    dd.size = shape[0];
    dd.num = 1;
    return MGH_SUCCESS;
  }
  dd.decomposed = true;
  if (dd.method == MGH_DD_MAXDIM) {  // :209-236
    uint64_t mx = 0;
    for (int d = 0; d < D; d++)
      if (shape[d] > mx) {
        mx = shape[d];
        dd.dim = d;
      }
    std::vector<uint64_t> cs = dd.shape;
    bool prefetch = false;
    while (need(cs, prefetch)) {
      if (cs[dd.dim] <= 3) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "domain decomposition: not enough device memory");
      cs[dd.dim] = (cs[dd.dim] - 1) / 2 + 1;
      prefetch = (shape[dd.dim] - 1) / cs[dd.dim] + 1 > 1;
    }
    dd.size = cs[dd.dim];
    dd.num = (shape[dd.dim] - 1) / dd.size + 1;
  } else if (dd.method == MGH_DD_BLOCK) {  // :238-263, 335-349
    dd.size = cfg.block_size;
    if (dd.size < 3) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "block_size");
    for (;;) {
      std::vector<uint64_t> cs(D, dd.size);
      uint64_t cnt = 1;
      for (int d = 0; d < D; d++) cnt *= (shape[d] - 1) / dd.size + 1;
      if (!need(cs, cnt > 1)) break;
      if (dd.size <= 3) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "domain decomposition: not enough device memory");
      dd.size = (dd.size - 1) / 2 + 1;
    }
    dd.num = 1;
    for (int d = 0; d < D; d++) dd.num *= (shape[d] - 1) / dd.size + 1;
  } else if (dd.method == MGH_DD_VARIABLE) {  // :350-357
    if (cfg.domain_decomposition_dim < 0 || cfg.domain_decomposition_dim >= D ||
        !cfg.domain_decomposition_sizes || !cfg.num_domain_decomposition_sizes)
      return hl_fail(MGH_ERR_INVALID_ARGUMENT, "Variable domain decomposition needs dim and sizes");
    dd.dim = cfg.domain_decomposition_dim;
    dd.var_sizes.assign(cfg.domain_decomposition_sizes,
                        cfg.domain_decomposition_sizes + cfg.num_domain_decomposition_sizes);
    uint64_t sum = 0;
    for (uint64_t v : dd.var_sizes) sum += v;
    if (sum != shape[dd.dim]) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "Variable sizes do not add up to the extent");
    dd.num = dd.var_sizes.size();
    dd.size = dd.var_sizes[0];
  } else {
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "domain_decomposition");
  }
  for (uint64_t id = 0; id < dd.num; id++)
    for (uint64_t e : dd.subdomain_shape(id))
      if (e < 3) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "domain decomposition leaves a subdomain with fewer than 3 nodes in a dimension");
  return MGH_SUCCESS;
}

// ---- host side of the host <-> device transfers ----------------------------------------------
// Numbers of the box the design follows (tools/micro/host_link.hip, 512 MB): pinned DMA 57.6 GB/s
// either way in one piece, 56.7 in 64 MB pieces, 55.2 in 16 MB pieces; memcpy pageable <-> pinned
// 32 GB/s on one thread, 85 on 4, 120 on 8; a ring of four pinned 16-32 MB slots filled by 4
// persistent threads while the DMA drains them: 52-53 GB/s; first touch of 512 MB of fresh 4 KB
// pages 72 ms on one thread (40 ms inside hipMemcpy), 3 ms with transparent huge pages and 8 threads;
// hipHostRegister of 512 MB 16 ms.
//
// Copy pool: a few persistent threads of the calling thread (released with its cache). A job is
// a function over part numbers; whoever waits for a job works on it too, so a job always
// completes even when the workers are busy with an earlier one. (Round 5 spawned 7 std::threads
// per 32 MB chunk.)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
// CPUs next to the current device (local_cpulist of its PCI function), empty when sysfs does not say.
// The boxes seen have two sockets and the GPU hangs on one of them: copy threads on the other
// socket write the pinned slots across the socket link (512^3 f32 host to host, pageable, whole
// process on the GPU's node 14.9 / 14.4 ms, on the other node 15.7 / 16.5 ms: tools/exp_numa.sh).
inline std::vector<int> device_local_cpus() {
  std::vector<int> cpus;
  int dev = 0;
  char bus[64] = {0};
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) {
    (void)hipGetLastError();
    return cpus;
  }
  for (char *q = bus; *q; q++) *q = (char)std::tolower((unsigned char)*q);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
  std::FILE *f = std::fopen(path.c_str(), "r");
  if (!f) return cpus;
  char line[4096] = {0};
  if (std::fgets(line, sizeof line, f)) {
    for (char *tok = std::strtok(line, ",\n"); tok; tok = std::strtok(nullptr, ",\n")) {
      int a = 0, b = 0;
      if (std::sscanf(tok, "%d-%d", &a, &b) == 2) {
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) cpus.push_back(c);
      } else if (std::sscanf(tok, "%d", &a) == 1 && a < CPU_SETSIZE) {
        cpus.push_back(a);
      }
    }
  }
  std::fclose(f);
  return cpus;
}

struct HostPool {
  static constexpr int kMaxWorkers = 31;
  std::vector<int> near_cpus;  // where the workers run (MGH_HL_COPY_AFFINITY=0: wherever the scheduler puts them)
  int kWorkers = (int)env_get("MGH_HL_COPY_THREADS", 5) - 1;  // copy threads beside the calling one
  struct Job {
    std::function<void(int)> fn;
    int nparts = 0;
    std::atomic<int> next{0}, done{0};
  };
  std::thread th[kMaxWorkers];
  bool started = false;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<uint64_t> gen{0};
  std::shared_ptr<Job> cur;  // (guarded by mu)
  bool stop = false;

  static void work(Job &j) {
    for (;;) {
      const int p = j.next.fetch_add(1, std::memory_order_relaxed);
      if (p >= j.nparts) break;
      j.fn(p);
      j.done.fetch_add(1, std::memory_order_release);
    }
  }
  void worker() {
    if (!near_cpus.empty()) {
      cpu_set_t set;
      CPU_ZERO(&set);
      for (int c : near_cpus) CPU_SET(c, &set);
      (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);  // (refused by a cpuset: stay where we are)
    }
    uint64_t seen = 0;
    for (;;) {
      std::shared_ptr<Job> j;
      // a chunked transfer posts a job every few hundred microseconds: poll that long before sleeping
      for (int spin = 0; spin < 4000 && gen.load(std::memory_order_acquire) == seen; spin++) cpu_relax();
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || gen.load(std::memory_order_acquire) != seen; });
        if (stop) return;
        seen = gen.load(std::memory_order_acquire);
        j = cur;
      }
      if (j) work(*j);
    }
  }
  std::shared_ptr<Job> post(std::function<void(int)> fn, int nparts) {
    auto j = std::make_shared<Job>();
    j->fn = std::move(fn);
    j->nparts = nparts;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!started) {
        if (env_get("MGH_HL_COPY_AFFINITY", 1) != 0) near_cpus = device_local_cpus();
        for (int k = 0; k < kWorkers; k++) th[k] = std::thread([this] { worker(); });
        started = true;
      }
      cur = j;
      gen.fetch_add(1, std::memory_order_release);
    }
    cv.notify_all();
    return j;
  }
  static void wait(Job &j) {
    work(j);
    while (j.done.load(std::memory_order_acquire) < j.nparts) cpu_relax();
  }
  void run(std::function<void(int)> fn, int nparts) {
    auto j = post(std::move(fn), nparts);
    wait(*j);
  }
  // dst <- src on the pool (both host memory)
  void copy(void *dst, const void *src, size_t bytes) {
    if (bytes < ((size_t)1 << 20)) {
      std::memcpy(dst, src, bytes);
      return;
    }
    // (one part per thread. Several smaller parts per thread, taken as the threads get to them, were
    // meant to balance a slow thread and LOSE: 512^3 f32 pageable, process on the far socket, 15.0 /
    // 14.5 ms with 1 part per thread, 17-22 / 15-17 ms with 4, 17.6 / 16 ms with 8; near socket 15.0 /
    // 14.4 against 15.4 / 14.7 and 15.0 / 14.6 -- MGH_HL_COPY_PARTS)
    static const size_t per_thread = (size_t)env_get("MGH_HL_COPY_PARTS", 1);
    const int kParts = (int)std::max<size_t>(1, std::min<size_t>(per_thread * (kWorkers + 1), bytes >> 18));
    const size_t part = (bytes / kParts + 4095) / 4096 * 4096;
    run([=](int t) {
      const size_t lo = std::min(bytes, (size_t)t * part), hi = std::min(bytes, (size_t)(t + 1) * part);
      if (hi > lo) std::memcpy((char *)dst + lo, (const char *)src + lo, hi - lo);
    }, kParts);
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!started) return;
      stop = true;
    }
    cv.notify_all();
    for (int k = 0; k < kWorkers; k++) th[k].join();
    started = false;
    stop = false;
    cur.reset();
  }
};
thread_local HostPool *g_pool_ptr = nullptr;
// The pool's worker threads end with the thread that owns them (a thread_local destructor: joining
// std::threads makes no HIP call, unlike the release of the cache and the pinned rings, which stays
// with mgh_release_cache) -- a caller thread that exits without releasing leaves memory behind, not
// threads.
struct HostPoolGuard {
  ~HostPoolGuard() {
    if (g_pool_ptr) {
      g_pool_ptr->shutdown();
      delete g_pool_ptr;
      g_pool_ptr = nullptr;
    }
  }
};
inline HostPool &host_pool() {
  static thread_local HostPoolGuard guard;
  (void)guard;
  if (!g_pool_ptr) g_pool_ptr = new HostPool();
  return *g_pool_ptr;
}

// Ring of pinned slots between PAGEABLE host memory and the device: the pool fills (drains) slot
// c % kSlots while the DMA engine works on the slots before it. (The reference registers the
// caller's buffers instead -- auto_pin_host_buffers -- at 16 ms per 512 MB on this box, and see
// mgh_config_default.) One ring per direction and thread, released with the cache.
struct PinnedRing {
  static constexpr int kSlots = 4;
  size_t kChunk = (size_t)env_get("MGH_HL_RING_MB", 16) << 20;
  void *buf[kSlots] = {};
  hipEvent_t ev[kSlots] = {};
  int dev = -1;  // device the events belong to
  int ensure(bool buffers) {
    int cur = 0;
    HL_HIP(hipGetDevice(&cur));
    if (cur != dev) {  // events recorded into another device's stream fail: recreate them
      release();
      dev = cur;
    }
    for (int i = 0; i < kSlots; i++) {
      if (buffers && !buf[i]) HL_HIP(hipHostMalloc(&buf[i], kChunk, hipHostMallocDefault));
      if (!ev[i]) HL_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    return MGH_SUCCESS;
  }
  void release() {
    for (int i = 0; i < kSlots; i++) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      buf[i] = nullptr;
      ev[i] = nullptr;
    }
  }
};
// One ring per direction: the host->device prefetch of subdomain id+1 and the device->host copy of
// record id run on different streams of the same thread and must not share slots.
thread_local PinnedRing *g_ring_ptr[2] = {nullptr, nullptr};
inline PinnedRing &ring(int dir) {
  if (!g_ring_ptr[dir]) g_ring_ptr[dir] = new PinnedRing();
  return *g_ring_ptr[dir];
}
void release_host_transfer_state() {
  for (auto *&r : g_ring_ptr)
    if (r) {
      r->release();
      delete r;
      r = nullptr;
    }
  if (g_pool_ptr) {
    g_pool_ptr->shutdown();
    delete g_pool_ptr;
    g_pool_ptr = nullptr;
  }
}

// Called for every piece of a host -> device transfer once its copy has been QUEUED on the
// transfer's stream: [off, off + nb) of the destination is complete when `landed` fires (the event
// is recorded again for a later piece: wait for it -- hipStreamWaitEvent -- before returning).

// dst (device) <- src (pageable host). The source is consumed when the call returns; the device
// side is complete in stream order.
int staged_h2d(void *dst, const void *src, size_t bytes, hipStream_t st, const ChunkFn *on_chunk) {
  PinnedRing &b = ring(0);
  HL_TRY(b.ensure(true));
  HostPool &pool = host_pool();
  size_t off = 0;
  for (int c = 0; off < bytes; c++) {
    const int i = c % PinnedRing::kSlots;
    const size_t nb = std::min(b.kChunk, bytes - off);
    HL_HIP(hipEventSynchronize(b.ev[i]));  // the previous transfer out of this slot is done
    pool.copy(b.buf[i], (const char *)src + off, nb);
    HL_HIP(hipMemcpyAsync((char *)dst + off, b.buf[i], nb, hipMemcpyHostToDevice, st));
    HL_HIP(hipEventRecord(b.ev[i], st));
    if (on_chunk) HL_TRY((*on_chunk)(off, nb, b.ev[i]));
    off += nb;
  }
  return MGH_SUCCESS;
}

// dst (pageable host) <- src (device), after everything queued on st. Complete on return.
int staged_d2h(void *dst, const void *src, size_t bytes, hipStream_t st) {
  PinnedRing &b = ring(1);
  HL_TRY(b.ensure(true));
  HostPool &pool = host_pool();
  constexpr int K = PinnedRing::kSlots;
  size_t off = 0, done = 0;
  size_t len[K] = {};
  int c = 0;
  for (; off < bytes; c++) {
    const int i = c % K;
    HL_HIP(hipEventSynchronize(b.ev[i]));  // (c < K: a transfer of an earlier call may still use the slot)
    if (c >= K) {  // drain the slot we are about to reuse
      pool.copy((char *)dst + done, b.buf[i], len[i]);
      done += len[i];
    }
    len[i] = std::min(b.kChunk, bytes - off);
    HL_HIP(hipMemcpyAsync(b.buf[i], (const char *)src + off, len[i], hipMemcpyDeviceToHost, st));
    HL_HIP(hipEventRecord(b.ev[i], st));
    off += len[i];
  }
  for (int k = std::max(0, c - K); k < c; k++) {
    const int i = k % K;
    HL_HIP(hipEventSynchronize(b.ev[i]));
    pool.copy((char *)dst + done, b.buf[i], len[i]);
    done += len[i];
  }
  return MGH_SUCCESS;
}

// contiguous copy between any two of device / pinned host / pageable host memory
int copy_any(void *dst, const void *src, size_t bytes, hipStream_t st, const ChunkFn *on_chunk) {
  constexpr size_t kStagedMin = (size_t)8 << 20;
  const bool dd = is_device_pointer(dst), sd = is_device_pointer(src);
  if (bytes >= kStagedMin) {
    if (dd && !sd && !is_registered_host(src)) return staged_h2d(dst, src, bytes, st, on_chunk);
    if (sd && !dd && !is_registered_host(dst)) return staged_d2h(dst, src, bytes, st);
  }
  if (on_chunk && dd && !sd) {
    // pinned source, a consumer per piece: 64 MB pieces (56.7 GB/s against 57.6 in one piece) for the
    // big transfers, 16 MB (55.2 GB/s) where a consumer that is slower than the link follows the
    // pieces (a record of a few hundred MB and its decoder)
    const size_t kPiece = bytes >= ((size_t)256 << 20) ? (size_t)64 << 20 : (size_t)16 << 20;
    PinnedRing &b = ring(0);
    HL_TRY(b.ensure(false));
    size_t off = 0;
    for (int c = 0; off < bytes; c++) {
      const size_t nb = std::min(kPiece, bytes - off);
      hipEvent_t ev = b.ev[c % PinnedRing::kSlots];
      HL_HIP(hipMemcpyAsync((char *)dst + off, (const char *)src + off, nb, hipMemcpyHostToDevice, st));
      HL_HIP(hipEventRecord(ev, st));
      HL_TRY((*on_chunk)(off, nb, ev));
      off += nb;
    }
    return MGH_SUCCESS;
  }
  HL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
  return MGH_SUCCESS;
}

// Host memory the library hands to the caller (released with free()): 2 MB aligned and advised for
// transparent huge pages -- the first touch of 512 MB costs 3 ms that way instead of 40-70 ms of
// 4 KB faults inside the device -> host copy. `touch`: fault the pages in now, on a few short-lived
// threads the caller joins before it writes (mgh_decompress: while the device decodes).
void *host_alloc_large(size_t bytes) {
  constexpr size_t kHuge = (size_t)2 << 20;
  if (bytes < 4 * kHuge) return std::malloc(bytes);
  void *p = nullptr;
  if (posix_memalign(&p, kHuge, bytes) != 0) return nullptr;
  (void)madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}
struct Pretouch {
  std::vector<std::thread> th;
  void start(void *p, size_t bytes) {
    constexpr int kThreads = 8;
    if (bytes < ((size_t)64 << 20)) return;
    const size_t part = (bytes / kThreads + 4095) / 4096 * 4096;
    for (int t = 0; t < kThreads; t++) {
      const size_t lo = std::min(bytes, (size_t)t * part), hi = std::min(bytes, (size_t)(t + 1) * part);
      if (hi > lo)
        th.emplace_back([=] {
          for (size_t o = lo; o < hi; o += 4096) ((volatile char *)p)[o] = 0;
        });
    }
  }
  void join() {
    for (auto &t : th) t.join();
    th.clear();
  }
  ~Pretouch() { join(); }
};

// A box of an array <-> a dense buffer, both in memory of the current device: one launch whatever
// the dimension (the reference copies subdomains with its own N-D kernels too:
// DomainDecomposer.hpp:649-845). One wave per row of the box; W = 4- or 8-byte words.
struct BoxCopy {
  uint32_t ext[MGH_MAX_DIM];      // extents of the box, leading 1s
  uint64_t fstride[MGH_MAX_DIM];  // element strides of the full array for those dims
  uint64_t rows;                  // product of all extents but the last
};
template <typename W>
__global__ void __launch_bounds__(256)
k_copy_box(W *__restrict__ dense, W *__restrict__ full, BoxCopy B, int to_dense) {
  const int lane = threadIdx.x & 63;
  for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < B.rows; row += (uint64_t)gridDim.x * 4) {
    uint64_t r = row, fo = 0;
#pragma unroll
    for (int d = MGH_MAX_DIM - 2; d >= 0; d--) {
      const uint64_t q = r / B.ext[d];
      fo += (r - q * B.ext[d]) * B.fstride[d];
      r = q;
    }
    W *f = full + fo;
    W *s = dense + row * B.ext[MGH_MAX_DIM - 1];
    if (to_dense)
      for (uint32_t k = lane; k < B.ext[MGH_MAX_DIM - 1]; k += 64) s[k] = f[k];
    else
      for (uint32_t k = lane; k < B.ext[MGH_MAX_DIM - 1]; k += 64) f[k] = s[k];
  }
}

// copy_subdomain (DomainDecomposer.hpp:649-845): dense subdomain buffer <-> its box inside the
// full array (host or device). Device to device on one GPU: ONE kernel launch. Otherwise as few
// strided copies as the box allows -- one hipMemcpy3DAsync per 3-D sub-box (round 5; a 2-D copy per
// r-plane was 129 calls of ~5 us for a 129^3 block, more than the block's compression).
int copy_subdomain(const Decomposer &dd, uint64_t id, size_t elem, void *sub, const void *full_c,
                   void *full_m, bool to_sub, hipStream_t st) {
  hl_debug(to_sub ? "copy_subdomain: to subdomain" : "copy_subdomain: to original");
  const int D = dd.D;
  const auto ext = dd.subdomain_shape(id), off = dd.subdomain_offset(id);
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *full_any = to_sub ? full_c : (const void *)full_m;
    if ((elem == 4 || elem == 8) && is_device_pointer_on(full_any, dev) && is_device_pointer_on(sub, dev)) {
      BoxCopy B{};
      uint64_t fs = 1, first = 0;
      std::vector<uint64_t> fstr(D);
      for (int d = D - 1; d >= 0; d--) {
        fstr[d] = fs;
        fs *= dd.shape[d];
      }
      B.rows = 1;
      for (int k = 0; k < MGH_MAX_DIM; k++) {
        const int d = k - (MGH_MAX_DIM - D);
        B.ext[k] = d >= 0 ? (uint32_t)ext[d] : 1u;
        B.fstride[k] = d >= 0 ? fstr[d] : 0;
        if (d >= 0) first += off[d] * fstr[d];
        if (k < MGH_MAX_DIM - 1) B.rows *= B.ext[k];
      }
      const unsigned grid = (unsigned)std::min<uint64_t>((B.rows + 3) / 4, 256 * 32);
      char *fp = (char *)const_cast<void *>(full_any) + first * elem;
      if (elem == 4)
        k_copy_box<uint32_t><<<grid, 256, 0, st>>>((uint32_t *)sub, (uint32_t *)fp, B, to_sub ? 1 : 0);
      else
        k_copy_box<uint64_t><<<grid, 256, 0, st>>>((uint64_t *)sub, (uint64_t *)fp, B, to_sub ? 1 : 0);
      HL_HIP(hipGetLastError());
      return MGH_SUCCESS;
    }
  }
  // merge trailing dimensions the box spans completely
  int k = D - 1;
  while (k > 0 && ext[k] == dd.shape[k]) k--;
  // width = ext[k] * prod(shape[k+1:]) contiguous elements; rows along dim k-1
  size_t inner = 1;
  for (int d = k + 1; d < D; d++) inner *= dd.shape[d];
  const size_t width = ext[k] * inner * elem;        // bytes per contiguous run
  const size_t full_pitch = dd.shape[k] * inner * elem;  // distance between runs (dim k-1)
  const size_t rows = k >= 1 ? ext[k - 1] : 1;
  // outer dims 0 .. k-2
  std::vector<uint64_t> idx(std::max(k - 1, 0), 0);
  std::vector<size_t> fstride(D);  // element strides of the full array
  {
    size_t s = 1;
    for (int d = D - 1; d >= 0; d--) {
      fstride[d] = s;
      s *= dd.shape[d];
    }
  }
  if (k == 0) {  // one contiguous run
    const size_t fo = off[0] * fstride[0];
    if (to_sub) return copy_any(sub, (const char *)full_c + fo * elem, width, st);
    return copy_any((char *)full_m + fo * elem, sub, width, st);
  }
  size_t sub_off = 0;
  // slabs of dim k-2 travel in one 3-D copy each (depth = ext[k-2]; the full array's slice pitch is
  // a whole number of its rows); the loop runs over the dims in front of it
  const bool use3d = k >= 2;
  const size_t depth = use3d ? ext[k - 2] : 1;
  const size_t sub_block = rows * width * depth;
  const int outer = use3d ? k - 2 : k - 1;  // dims 0 .. outer-1 are looped over
  for (;;) {
    size_t fo = off[k] * fstride[k];
    if (k >= 1) fo += off[k - 1] * fstride[k - 1];
    if (use3d) fo += off[k - 2] * fstride[k - 2];
    for (int d = 0; d < outer; d++) fo += (off[d] + idx[d]) * fstride[d];
    char *sp = (char *)sub + sub_off;
    char *fp = (char *)const_cast<void *>(to_sub ? full_c : (const void *)full_m) + fo * elem;
    if (use3d) {
      hipMemcpy3DParms pr{};
      // (pitched pointers: pitch in bytes, then the allocation's width in bytes and height in rows)
      const hipPitchedPtr dense = make_hipPitchedPtr(sp, width, width, rows);
      const hipPitchedPtr whole = make_hipPitchedPtr(fp, full_pitch, full_pitch, dd.shape[k - 1]);
      pr.srcPtr = to_sub ? whole : dense;
      pr.dstPtr = to_sub ? dense : whole;
      pr.extent = make_hipExtent(width, rows, depth);
      pr.kind = hipMemcpyDefault;
      HL_HIP(hipMemcpy3DAsync(&pr, st));
    } else if (to_sub) {
      HL_HIP(hipMemcpy2DAsync(sp, width, fp, full_pitch, width, rows, hipMemcpyDefault, st));
    } else {
      HL_HIP(hipMemcpy2DAsync(fp, full_pitch, sp, width, width, rows, hipMemcpyDefault, st));
    }
    sub_off += sub_block;
    int d = outer - 1;
    while (d >= 0) {
      if (++idx[d] < ext[d]) break;
      idx[d] = 0;
      d--;
    }
    if (d < 0) break;
  }
  return MGH_SUCCESS;
}

// ---- per-thread cache: hierarchies, device buffers, lossless context
// (CompressorCache, CompressionLowLevel/CompressorCache.hpp:139-142) -------------------------
// One compute lane of the subdomain pipeline: a stream with everything a subdomain needs between
// its decomposition and its record (quantized symbols, outlier lists, lossless context, pinned
// scratch). Two lanes: while subdomain k is in histogram -> code construction (host) -> encoder
// -> record on one lane, subdomain k+1 is decomposed and quantized on the other
// (GPUPipelines.hpp:88-207 runs 2 buffers x 3 queues the same way).
struct Lane {
  hipStream_t st = nullptr;
  DevBuf q, q2, ocount, oidx, oval;  // q2: level-linearised copy (config.reorder == 1)
  DevBuf sub;                        // decompression: the dense subdomain when the output is not written in place
  PinBuf pin;                        // [0, 8) size prefix of a raw record, [16, 24) norm read-back
  mgh_lossless_ctx *ll = nullptr;
  uint64_t ocap = 0;                 // elements oidx / oval hold
  void release() {
    q.release();
    q2.release();
    ocount.release();
    oidx.release();
    oval.release();
    sub.release();
    pin.release();
    if (ll) mgh_lossless_destroy(ll);
    ll = nullptr;
    ocap = 0;
    if (st) (void)hipStreamDestroy(st);
    st = nullptr;
  }
};
constexpr int kLanes = 2;
constexpr int kInBufs = 3;  // subdomain k in its record phase, k+1 being decomposed, k+2 arriving

struct HlCache {
  std::map<std::vector<uint64_t>, mgh_hierarchy *> hier;  // key: lane, dtype, normalize, max_level, shape...
  DevBuf in[kInBufs];
  hipEvent_t in_ready[kInBufs] = {nullptr, nullptr, nullptr};  // the copy into in[b] has landed
  hipEvent_t in_free[kInBufs] = {nullptr, nullptr, nullptr};   // the last reader of in[b] is done (norm phase)
  PinBuf hpin;  // pinned staging of mgh_compress: [64, ...) the header
  Lane lane[kLanes];
  hipStream_t copy_st = nullptr;  // prefetch of the next subdomains (compression), strided write-back (decompression)
  hipStream_t aux_st = nullptr;   // small device -> host reads that must not wait for the lanes
  PinBuf aux_pin;
  int dev = -1;
  // device bytes the cache holds now (they are reused, so they count as available to the next call)
  size_t held_bytes() const {
    size_t b = 0;
    for (auto &kv : hier) b += mgh_device_bytes(kv.second);
    for (auto &x : in) b += x.cap;
    for (auto &l : lane) {
      b += l.q.cap + l.q2.cap + l.oidx.cap + l.oval.cap + l.sub.cap;
      if (l.ll) b += l.ll->units.cap + l.ll->oidx.cap + l.ll->oval.cap;
    }
    return b;
  }
  void release() {
    for (auto &kv : hier) mgh_hierarchy_destroy(kv.second);
    hier.clear();
    for (auto &b : in) b.release();
    for (auto *arr : {in_ready, in_free})
      for (int b = 0; b < kInBufs; b++) {
        if (arr[b]) (void)hipEventDestroy(arr[b]);
        arr[b] = nullptr;
      }
    hpin.release();
    aux_pin.release();
    for (auto &l : lane) l.release();
    for (hipStream_t *s : {&copy_st, &aux_st}) {
      if (*s) (void)hipStreamDestroy(*s);
      *s = nullptr;
    }
    dev = -1;
  }
};
// Deliberately never destroyed automatically: a thread_local destructor would run HIP calls at
// thread / process exit, possibly after the HIP runtime has started to tear itself down. The
// resources are returned by mgh_release_cache() (like mgard_x::release_cache); a thread that
// exits without calling it leaves them to process teardown.
thread_local HlCache *g_cache_ptr = nullptr;
inline HlCache &hl_cache() {
  if (!g_cache_ptr) g_cache_ptr = new HlCache();
  return *g_cache_ptr;
}
#define g_cache (hl_cache())

hipStream_t cache_copy_stream(int dev) { return g_cache_ptr && g_cache_ptr->dev == dev ? g_cache_ptr->copy_st : nullptr; }

// Small synchronous device -> host read (record sizes, record heads). On the cache's own stream
// and through its pinned buffer: hipMemcpy() would run on the NULL stream and with it wait for
// everything the lanes have queued. Stand-alone calls that never prepared a cache use hipMemcpy.
int aux_read(void *dst, const void *src, size_t bytes) {
  HlCache *c = g_cache_ptr;
  if (!c || !c->aux_st || !c->aux_pin.p) {
    HL_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MGH_SUCCESS;
  }
  for (size_t off = 0; off < bytes; off += c->aux_pin.cap) {
    const size_t nb = std::min(c->aux_pin.cap, bytes - off);
    HL_HIP(hipMemcpyAsync(c->aux_pin.p, (const char *)src + off, nb, hipMemcpyDeviceToHost, c->aux_st));
    HL_HIP(hipStreamSynchronize(c->aux_st));
    std::memcpy((char *)dst + off, c->aux_pin.p, nb);
  }
  return MGH_SUCCESS;
}

int cache_prepare(int dev) {
  if (g_cache.dev != dev) {
    g_cache.release();
    HL_HIP(hipSetDevice(dev));
    // Default (blocking) flags like the reference's queues (DeviceAdapterHip.h:514): the pipeline
    // streams are ordered against the NULL stream, so a device-resident input that an earlier
    // kernel / copy on the NULL stream (torch's default stream) is still producing is complete
    // before the first pipeline stage reads it, and work the caller queues on the NULL stream
    // afterwards waits for the pipeline. Inputs produced on OTHER non-blocking streams must be
    // synchronised by the caller (include/mgard_hip_compress.h).
    // The two lanes must sit on DIFFERENT hardware queues, or the pipeline is a sequence again. The
    // runtime multiplexes all streams of one priority over a few hardware queues (GPU_MAX_HW_QUEUES,
    // 4 by default) by use count: in a process that already holds many streams -- torch keeps a pool
    // of 32 -- two freshly created ones can share a queue (seen in bench.py: both lanes on one queue,
    // 64 x 512^3 decompression 97 instead of 86 ms). Queues are pooled PER PRIORITY, so the lanes get
    // different priorities (they swap roles with every subdomain, neither is favoured for long), and
    // the copy / small-read streams the third level, away from both.
    int prio_least = 0, prio_greatest = 0;
    HL_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const bool three = prio_least > 0 && prio_greatest < 0;
    for (int l = 0; l < kLanes; l++) {
      HL_HIP(hipStreamCreateWithPriority(&g_cache.lane[l].st, hipStreamDefault, l == 0 ? 0 : prio_greatest));
      HL_TRY(mgh_lossless_create(&g_cache.lane[l].ll, dev));
    }
    HL_HIP(hipStreamCreateWithPriority(&g_cache.copy_st, hipStreamDefault, three ? prio_least : 0));
    HL_HIP(hipStreamCreateWithPriority(&g_cache.aux_st, hipStreamDefault, three ? prio_least : 0));
    for (auto &e : g_cache.in_ready) HL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : g_cache.in_free) HL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HL_TRY(g_cache.aux_pin.ensure(512 * 1024));
    g_cache.dev = dev;
  }
  return MGH_SUCCESS;
}

// Hierarchy of a subdomain (DomainDecomposer::subdomain_hierarchy, :265-301). Uniform ones
// are cached by shape (at most 4 alive); non-uniform ones are built per subdomain.
constexpr size_t kHierCacheMax = 8 * kLanes;  // (a Block decomposition of a 3-D array has up to 8 subdomain shapes)
// start of a high-level call (nothing in flight): make room when the cache is full
int trim_hierarchy_cache() {
  if (g_cache.hier.size() < kHierCacheMax) return MGH_SUCCESS;
  for (auto &kv : g_cache.hier) mgh_hierarchy_destroy(kv.second);
  g_cache.hier.clear();
  return MGH_SUCCESS;
}

// A hierarchy owns the workspace of the subdomain that runs on it, so the two lanes of the
// pipeline keep separate ones (`lane` is part of the key).
int get_hierarchy(mgh_hierarchy **out, bool *owned, int dtype, const std::vector<uint64_t> &shape,
                  const std::vector<std::vector<double>> *coords, const std::vector<uint64_t> &off,
                  const mgh_config &cfg, int lane = 0) {
  const int D = (int)shape.size();
  if (!coords) {
    std::vector<uint64_t> key = {(uint64_t)lane, (uint64_t)dtype, (uint64_t)cfg.normalize_coordinates,
                                 cfg.max_larget_level};
    key.insert(key.end(), shape.begin(), shape.end());
    auto it = g_cache.hier.find(key);
    if (it != g_cache.hier.end()) {
      *out = it->second;
      *owned = false;
      return MGH_SUCCESS;
    }
    mgh_hierarchy *h = nullptr;
    HL_TRY(mgh_hierarchy_create(&h, D, shape.data(), dtype, nullptr, cfg.normalize_coordinates,
                                cfg.max_larget_level, cfg.dev_id));
    *out = h;
    // A full cache is emptied between calls only (trim_hierarchy_cache): inside a call a cached
    // hierarchy may be at work on the other lane. A shape that does not fit any more is built for
    // its subdomain alone.
    *owned = g_cache.hier.size() >= kHierCacheMax;
    if (!*owned) g_cache.hier[key] = h;
    return MGH_SUCCESS;
  }
  // slice of the coordinate arrays, converted to the data type
  std::vector<std::vector<float>> cf(D);
  std::vector<std::vector<double>> cd(D);
  const void *ptrs[MGH_MAX_DIM];
  for (int d = 0; d < D; d++) {
    const double *src = (*coords)[d].data() + off[d];
    if (dtype == MGH_FLOAT) {
      cf[d].assign(src, src + shape[d]);
      ptrs[d] = cf[d].data();
    } else {
      cd[d].assign(src, src + shape[d]);
      ptrs[d] = cd[d].data();
    }
  }
  HL_TRY(mgh_hierarchy_create(out, D, shape.data(), dtype, ptrs, cfg.normalize_coordinates,
                              cfg.max_larget_level, cfg.dev_id));
  *owned = true;
  return MGH_SUCCESS;
}

// calc_local_abs_tol (ErrorToleranceCalculator.hpp:134-155), in the data type
template <typename T> T local_abs_tol(int ebtype, T norm, T tol, T s, uint64_t nsub) {
  if (ebtype == MGH_REL) {
    if (s == std::numeric_limits<T>::infinity()) return tol * norm;
    return std::sqrt((tol * norm) * (tol * norm) / (T)nsub);
  }
  if (s == std::numeric_limits<T>::infinity()) return tol;
  return std::sqrt((tol * tol) / (T)nsub);
}

int header_from(const Decomposer &dd, int dtype, int ebtype, double tol, double s, double norm,
                const std::vector<std::vector<double>> *coords, const mgh_config &cfg, fmt::Header &h) {
  h = fmt::Header();
  h.is_double = dtype == MGH_DOUBLE;
  h.shape = dd.shape;
  h.uniform = coords == nullptr;
  if (coords) h.coords = *coords;
  h.rel = ebtype == MGH_REL;
  h.tol = tol;
  h.s = s;
  h.norm = norm;
  if (dd.decomposed) {
    h.dd_method = dd.method == MGH_DD_MAXDIM ? fmt::DD_MAX_DIMENSION
                  : dd.method == MGH_DD_BLOCK ? fmt::DD_BLOCK : fmt::DD_VARIABLE;
  } else {
    h.dd_method = fmt::DD_NOOP;
  }
  h.dd_dim = dd.dim;
  h.dd_size = dd.size;
  h.hierarchy = fmt::HIER_MULTIDIM;
  h.l_target = 0;  // never filled by the reference (Metadata.hpp:78-113)
  h.reorder = cfg.reorder != 0;
  h.compressor = cfg.lossless == MGH_LOSSLESS_HUFFMAN ? fmt::COMP_X_HUFFMAN : fmt::COMP_X_HUFFMAN_ZSTD;
  h.huff_dict_size = cfg.huff_dict_size;
  h.huff_block_size = cfg.huff_block_size;
  h.backend = fmt::DEV_X_HIP;
  return MGH_SUCCESS;
}

template <typename T>
int compress_impl(int D, int dtype, const uint64_t *shape, double tol_d, double s_d, int ebtype,
                  const void *original, void **compressed, size_t *compressed_size,
                  const void *const *coords_in, const mgh_config &cfg, bool prealloc) {
  HL_TRY(cache_prepare(cfg.dev_id));
  HL_TRY(trim_hierarchy_cache());
  const T tol = (T)tol_d, s = (T)s_d;
  size_t total = 1;
  for (int d = 0; d < D; d++) total *= shape[d];
  const size_t elem = sizeof(T);
  const bool in_dev = is_device_pointer(original);
  Decomposer dd;
  HL_TRY(make_decomposer(dd, D, shape, elem, cfg));
  std::vector<std::vector<double>> coords;
  if (coords_in) {
    coords.resize(D);
    for (int d = 0; d < D; d++) {
      const T *c = static_cast<const T *>(coords_in[d]);
      coords[d].assign(c, c + shape[d]);
    }
  }
  const std::vector<std::vector<double>> *cptr = coords_in ? &coords : nullptr;
  // output buffer, same memory space as the input (CompressionHighLevel.hpp:147-162)
  size_t cap;
  if (!prealloc) {
    cap = total * elem + (size_t)1e6;
    if (in_dev) HL_HIP(hipMalloc(compressed, cap));
    else if (!(*compressed = host_alloc_large(cap))) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "malloc");
  } else {
    cap = *compressed_size;
  }
  const bool out_dev = is_device_pointer(*compressed);
  // pin the host buffers for asynchronous transfers (auto_pin_host_buffers,
  // CompressionHighLevel.hpp:164-189: input and output)
  bool pinned_here = false, out_pinned_here = false;
  if (!in_dev && cfg.auto_pin_host_buffers && !is_registered_host(original)) {
    if (hipHostRegister(const_cast<void *>(original), total * elem, hipHostRegisterDefault) == hipSuccess)
      pinned_here = true;
    else
      (void)hipGetLastError();
  }
  if (!out_dev && cfg.auto_pin_host_buffers && !is_registered_host(*compressed)) {
    if (hipHostRegister(*compressed, cap, hipHostRegisterDefault) == hipSuccess)
      out_pinned_here = true;
    else
      (void)hipGetLastError();
  }
  // MGH_HL_PIPELINE=0: every subdomain runs start to end before the next one is queued (cross-check;
  // same container byte for byte up to the order of the outlier lists)
  const bool pipelined = env_get("MGH_HL_PIPELINE", 1) != 0;
  int nlanes = dd.num > 1 && pipelined ? kLanes : 1;
  auto drain = [&] {  // nothing of this call may be in flight when it returns
    for (int l = 0; l < kLanes; l++) (void)hipStreamSynchronize(g_cache.lane[l].st);
    (void)hipStreamSynchronize(g_cache.copy_st);
  };
  std::vector<mgh_hierarchy *> owned_alive;  // per-subdomain hierarchies (non-uniform grids) not yet destroyed
  auto cleanup = [&](int rc) {
    if (rc != MGH_SUCCESS) drain();
    for (mgh_hierarchy *h : owned_alive) mgh_hierarchy_destroy(h);
    owned_alive.clear();
    if (pinned_here) (void)hipHostUnregister(const_cast<void *>(original));
    if (out_pinned_here) (void)hipHostUnregister(*compressed);
    if (rc != MGH_SUCCESS && !prealloc) {
      if (in_dev) (void)hipFree(*compressed); else std::free(*compressed);
      *compressed = nullptr;
    }
    return rc;
  };
  const uint64_t max_elems = dd.max_subdomain_elems();
  const uint64_t ocap = std::max<uint64_t>(1, (uint64_t)(cfg.estimate_outlier_ratio * (double)max_elems));
  // device-resident input whose subdomains are contiguous slabs: compress them where they are
  const bool zero_copy = in_dev && dd.all_contiguous();
  int nbufs = zero_copy ? 0 : (int)std::min<uint64_t>(dd.num, nlanes > 1 ? kInBufs : 2);
  // The subdomain sizes follow the REFERENCE's footprint estimate (make_decomposer). What this
  // implementation keeps resident is different -- less per subdomain, but the two-lane pipeline
  // holds up to three inputs and two sets of everything else: where that set does not fit what is
  // free (plus what the cache already holds), the schedule falls back to one lane and two inputs
  // instead of failing in an allocation half way through.
  if (nlanes > 1) {
    size_t free_b = 0, total_b = 0;
    HL_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t avail = std::min<size_t>(free_b + g_cache.held_bytes(), cfg.max_memory_footprint);
    if (own_resident_bytes(D, max_elems, elem, cfg, ocap, nlanes, nbufs) > avail) {
      nlanes = 1;
      nbufs = zero_copy ? 0 : (int)std::min<uint64_t>(dd.num, 2);
    }
  }
  auto sub_in = [&](uint64_t id) -> const void * {
    return zero_copy ? (const void *)((const char *)original + dd.linear_offset(id) * elem)
                     : (const void *)g_cache.in[id % nbufs].p;
  };
  // subdomain id into its input buffer on the copy stream; in_ready[buffer] fires when it has landed
  auto fetch_sub = [&](uint64_t id) -> int {
    if (zero_copy) return MGH_SUCCESS;
    const int b = (int)(id % nbufs);
    HL_TRY(copy_subdomain(dd, id, elem, g_cache.in[b].p, original, nullptr, true, g_cache.copy_st));
    HL_HIP(hipEventRecord(g_cache.in_ready[b], g_cache.copy_st));
    return MGH_SUCCESS;
  };
  int rc;
  auto ensure_all = [&]() -> int {
    for (int b = 0; b < nbufs; b++) HL_TRY(g_cache.in[b].ensure(max_elems * elem));
    for (int l = 0; l < nlanes; l++) {
      Lane &L = g_cache.lane[l];
      HL_TRY(L.q.ensure(max_elems * 8));
      if (cfg.reorder) HL_TRY(L.q2.ensure(max_elems * 8));
      HL_TRY(L.ocount.ensure(8));
      HL_TRY(L.oidx.ensure(ocap * 8));
      HL_TRY(L.oval.ensure(ocap * 8));
      L.ocap = std::max(L.ocap, ocap);  // (grow-only buffers: an earlier call may have left more)
      HL_TRY(L.pin.ensure(64));
    }
    return MGH_SUCCESS;
  };
  if ((rc = ensure_all()) != MGH_SUCCESS) return cleanup(rc);

  // norm of the whole domain when it is decomposed (calc_norm_decomposed_w_prefetch)
  T norm = 1;
  T local_tol = tol;
  int local_eb = ebtype;
  if (dd.decomposed) {
    if (ebtype == MGH_REL) {
      // every subdomain's norm is left on the device (one slot each) and all of them come back in
      // one copy: no host round trip between the reductions
      hipStream_t st = g_cache.lane[0].st;
      DevBuf &slots = g_cache.lane[0].q;  // (free until the pipeline starts)
      if ((rc = g_cache.aux_pin.ensure(std::max<size_t>(512 * 1024, dd.num * sizeof(T)))) != MGH_SUCCESS)
        return cleanup(rc);
      if ((rc = slots.ensure(dd.num * sizeof(T))) != MGH_SUCCESS) return cleanup(rc);
      for (uint64_t id = 0; id < dd.num; id++) {
        if (!zero_copy) {
          // (the buffers in turn: the copy of id waits for the reduction that read id - nbufs)
          if (id >= (uint64_t)nbufs && hipStreamWaitEvent(g_cache.copy_st, g_cache.in_free[id % nbufs], 0) != hipSuccess)
            return cleanup(hl_fail(MGH_ERR_DEVICE, "hipStreamWaitEvent"));
          if ((rc = fetch_sub(id)) != MGH_SUCCESS) return cleanup(rc);
          if (hipStreamWaitEvent(st, g_cache.in_ready[id % nbufs], 0) != hipSuccess)
            return cleanup(hl_fail(MGH_ERR_DEVICE, "hipStreamWaitEvent"));
        }
        mgh_hierarchy *h = nullptr;
        bool owned = false;
        const auto sshape = dd.subdomain_shape(id);
        if ((rc = get_hierarchy(&h, &owned, dtype, sshape, nullptr, dd.subdomain_offset(id), cfg, 0)) != MGH_SUCCESS)
          return cleanup(rc);
        if (owned) owned_alive.push_back(h);
        rc = mgh_norm_device(h, sub_in(id), s_d, (char *)slots.p + id * sizeof(T), st);
        if (rc != MGH_SUCCESS) return cleanup(rc);
        if (!zero_copy && hipEventRecord(g_cache.in_free[id % nbufs], st) != hipSuccess)
          return cleanup(hl_fail(MGH_ERR_DEVICE, "hipEventRecord"));
      }
      if (hipMemcpyAsync(g_cache.aux_pin.p, slots.p, dd.num * sizeof(T), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        return cleanup(hl_fail(MGH_ERR_DEVICE, "norms of the subdomains"));
      for (mgh_hierarchy *h : owned_alive) mgh_hierarchy_destroy(h);
      owned_alive.clear();
      double acc = 0;
      for (uint64_t id = 0; id < dd.num; id++) {
        const double ln = (double)((const T *)g_cache.aux_pin.p)[id];
        uint64_t cnt = 1;
        for (uint64_t e : dd.subdomain_shape(id)) cnt *= e;
        if (s == std::numeric_limits<T>::infinity()) acc = std::max(acc, ln);
        else acc += ln * ln * (cfg.normalize_coordinates ? (double)cnt : 1.0);  // un-normalised square
      }
      if (s == std::numeric_limits<T>::infinity()) norm = (T)acc;
      else norm = (T)(cfg.normalize_coordinates ? std::sqrt(acc / (double)total) : std::sqrt(acc));
    }
    local_tol = local_abs_tol<T>(ebtype, norm, tol, s, dd.num);
    local_eb = MGH_ABS;  // CompressionHighLevel.hpp:135-138
  }

  // metadata size is known up front: the header does not depend on the payloads, except for
  // the norm of a non-decomposed REL run, whose encoding has a fixed length (a non-zero double)
  fmt::Header hdr;
  header_from(dd, dtype, ebtype, tol_d, s_d, ebtype == MGH_REL ? (dd.decomposed ? (double)norm : 1.0) : 0.0, cptr, cfg, hdr);
  std::vector<uint8_t> meta;  // (outlives the asynchronous copy of the header below)
  const size_t meta_size = fmt::serialize_metadata(hdr).size();
  if (meta_size > cap) return cleanup(hl_fail(MGH_ERR_OUTPUT_TOO_LARGE, "output buffer too small for the header"));
  // (sized once, here: an asynchronous copy out of it must never meet a re-allocation)
  if ((rc = g_cache.hpin.ensure(64 + meta_size)) != MGH_SUCCESS) return cleanup(rc);
  size_t byte_offset = meta_size;

  // ---- subdomain pipeline (compress_pipeline_gpu, GPUPipelines.hpp:69-207) ----
  // issue(k): decomposition + quantizer + histogram of subdomain k queued on lane k % nlanes, no
  // host synchronisation; finish(k): code construction, encoder, record -- the host waits for lane
  // k % nlanes only. With two lanes issue(k + 1) is queued BEFORE finish(k), so the device has the
  // next subdomain's level passes and solves to run while this one's histogram travels to the
  // host and its encoder runs; the input of k + 2 arrives on the copy stream meanwhile.
  struct SubJob {
    uint64_t id = 0, n = 0;
    int lane = 0;
    mgh_hierarchy *h = nullptr;
    bool owned = false, sym16 = false, norm_deferred = false;
    double norm_out = 0;
    LosslessJob lj;
  };
  // (pre_h: the hierarchy of a subdomain whose norm was reduced while it arrived, see below)
  mgh_hierarchy *pre_h = nullptr;
  bool pre_owned = false;
  auto issue = [&](SubJob &J, uint64_t id) -> int {
    J = SubJob();
    J.id = id;
    J.lane = (int)(id % nlanes);
    Lane &L = g_cache.lane[J.lane];
    hipStream_t st = L.st;
    const auto sshape = dd.subdomain_shape(id);
    J.n = 1;
    for (uint64_t e : sshape) J.n *= e;
    if (!zero_copy) HL_HIP(hipStreamWaitEvent(st, g_cache.in_ready[id % nbufs], 0));
    if (pre_h) {
      J.h = pre_h;
      J.owned = pre_owned;
      pre_h = nullptr;
    } else {
      HL_TRY(get_hierarchy(&J.h, &J.owned, dtype, sshape, cptr, dd.subdomain_offset(id), cfg, J.lane));
      if (J.owned) owned_alive.push_back(J.h);
    }
    hl_debug("compress: subdomain ready");
    // (outlier counter of the lane + the lossless stage's bins and chunk states: one launch, here)
    if (cfg.reorder || cfg.lossless == MGH_LOSSLESS_HUFFMAN_ZSTD) {
      HL_HIP(hipMemsetAsync(L.ocount.p, 0, 8, st));
    } else {
      HL_TRY(lossless_prepare(L.ll, J.n, cfg.huff_dict_size, cfg.huff_block_size, st, (unsigned long long *)L.ocount.p));
      if (!L.ll->prepared_n) HL_HIP(hipMemsetAsync(L.ocount.p, 0, 8, st));
    }
    J.norm_out = (double)norm;
    // 16-bit symbols straight from the quantizer where the fused kernels run (a quarter of the
    // bytes written by the quantizer and read twice by the lossless stage); int64 otherwise
    if (!cfg.reorder && lossless_sym16_ok(cfg.huff_dict_size, cfg.huff_block_size)) {
      const int r16 = mgh_decompose_quantize_sym16(
          J.h, sub_in(id), local_eb, (double)local_tol, s_d, local_eb == MGH_REL ? 0.0 : (double)norm,
          nullptr /* the norm stays on the device: see below */, cfg.huff_dict_size, (uint16_t *)L.q.p,
          (uint64_t *)L.ocount.p, (uint64_t *)L.oidx.p, (int64_t *)L.oval.p, L.ocap, st);
      if (r16 == MGH_SUCCESS) J.sym16 = true;
      else if (r16 != MGH_ERR_UNSUPPORTED_DIMENSION) return r16;
    }
    if (!J.sym16)
      HL_TRY(mgh_decompose_quantize(J.h, sub_in(id), local_eb, (double)local_tol, s_d,
                                    local_eb == MGH_REL ? 0.0 : (double)norm,
                                    local_eb == MGH_REL ? &J.norm_out : nullptr, cfg.huff_dict_size, 1,
                                    (int64_t *)L.q.p, (uint64_t *)L.ocount.p, (uint64_t *)L.oidx.p,
                                    (int64_t *)L.oval.p, L.ocap, nullptr, st));
    hl_debug("compress: decompose + quantize done");
    // REL on the fused device path (the 16-bit symbol call: the norm never left the device): it
    // is fetched behind the quantizer into pinned memory and read when the lossless stage has
    // synchronised anyway -- a read-back inside the call above would stall the queue in front of
    // the histogram kernel. Every other path hands the norm back itself (norm_out).
    if (local_eb == MGH_REL && J.sym16) {
      J.norm_deferred = true;
      HL_HIP(hipMemcpyAsync((char *)L.pin.p + 16, mgh_norm_device_ptr(J.h), sizeof(T), hipMemcpyDeviceToHost, st));
    }
    const int64_t *q_enc = (const int64_t *)L.q.p;
    if (cfg.reorder) {
      // config.reorder == 1: the lossless stage sees the integers level by level, outlier
      // indices are positions in that array (LinearQuantization.hpp:226-232, 588-605)
      HL_TRY(mgh_level_linearize(J.h, (const int64_t *)L.q.p, (int64_t *)L.q2.p, 0, (uint64_t *)L.oidx.p,
                                 (const uint64_t *)L.ocount.p, 0, L.ocap, st));
      q_enc = (const int64_t *)L.q2.p;
    }
    LosslessJob &lj = J.lj;
    lj.d_q = q_enc;
    lj.n = J.n;
    lj.dict = cfg.huff_dict_size;
    lj.chunk = cfg.huff_block_size;
    lj.lossless = cfg.lossless;
    lj.zstd_level = cfg.zstd_compress_level;
    lj.d_oidx = (const uint64_t *)L.oidx.p;
    lj.d_oval = (const int64_t *)L.oval.p;
    lj.ocount = 0;  // (read back together with the encoder's results)
    lj.d_ocount = (const uint64_t *)L.ocount.p;
    lj.ocap = L.ocap;
    lj.cap_units = J.n * elem / 8 + 1;
    lj.sym16 = J.sym16;
    return lossless_begin(L.ll, lj, st);
  };
  auto finish = [&](SubJob &J) -> int {
    Lane &L = g_cache.lane[J.lane];
    hipStream_t st = L.st;
    const uint64_t id = J.id, n = J.n;
    int frc = MGH_SUCCESS;
    for (int attempt = 0; attempt < 2; attempt++) {
      const bool direct_ok = out_dev && cap - byte_offset > 8;
      frc = lossless_finish(L.ll, J.lj, st, direct_ok ? (uint8_t *)*compressed + byte_offset + 8 : nullptr,
                            direct_ok ? cap - byte_offset - 8 : 0);
      if (frc != kOutlierOverflow) break;
      if (attempt == 1) return hl_fail(MGH_ERR_DEVICE, "outlier lists overflowed twice");
      // estimate_outlier_ratio was too optimistic: grow this lane's lists to what the subdomain
      // needs and quantize it again (LinearQuantization.hpp:621-676 re-launches the same way)
      const uint64_t need = L.ll->outliers_needed;
      HL_TRY(L.oidx.ensure(need * 8));
      HL_TRY(L.oval.ensure(need * 8));
      L.ocap = need;
      mgh_hierarchy *h = J.h;
      const bool owned = J.owned;
      SubJob again;
      // (the hierarchy of the first attempt serves the second; issue() looks it up again)
      if (owned) {
        owned_alive.erase(std::remove(owned_alive.begin(), owned_alive.end(), h), owned_alive.end());
        mgh_hierarchy_destroy(h);
      }
      HL_TRY(issue(again, id));
      J = again;
    }
    HL_TRY(frc);
    if (J.norm_deferred) {  // (lossless_finish has synchronised the lane)
      T nv;
      std::memcpy(&nv, (const char *)L.pin.p + 16, sizeof(T));
      norm = nv;
    } else if (local_eb == MGH_REL) {
      norm = (T)J.norm_out;
    }
    hl_debug("compress: lossless stage done");
    uint64_t csize = L.ll->record_size();
    const bool raw = (double)(n * elem) / (double)csize < 1.0;  // GPUPipelines.hpp:136-155
    if (raw) csize = n * elem;
    if (csize > cap - byte_offset || cap - byte_offset - csize < 8)
      return hl_fail(MGH_ERR_OUTPUT_TOO_LARGE, "Output too large");
    char *dst = (char *)*compressed + byte_offset;
    const bool prefix_with_record = out_dev && !raw;  // (one copy: size prefix + head of the record)
    if (out_dev && raw) {
      std::memcpy(L.pin.p, &csize, 8);
      HL_HIP(hipMemcpyAsync(dst, L.pin.p, 8, hipMemcpyHostToDevice, st));
    } else if (!out_dev) {
      std::memcpy(dst, &csize, 8);
    }
    dst += 8;
    if (raw) {
      // the dense subdomain itself (decompose_quantize does not modify its input, so the buffer
      // still holds it)
      HL_TRY(copy_any(dst, sub_in(id), csize, st));
    } else {
      const uint64_t cs64 = csize;
      HL_TRY(record_write(L.ll, dst, st, prefix_with_record ? &cs64 : nullptr));
    }
    if (id + 1 == dd.num) {
      // header (with the norm the pipeline computed for a non-decomposed REL run): it travels on the
      // last record's stream, in front of the synchronisation that record needs anyway
      header_from(dd, dtype, ebtype, tol_d, s_d, ebtype == MGH_REL ? (double)norm : 0.0, cptr, cfg, hdr);
      meta = fmt::serialize_metadata(hdr);
      if (meta.size() != meta_size) return hl_fail(MGH_ERR_FORMAT, "metadata size changed");
      if (out_dev) {
        // (out of pinned memory: the copy is queued, not staged synchronously)
        std::memcpy((char *)g_cache.hpin.p + 64, meta.data(), meta.size());
        HL_HIP(hipMemcpyAsync(*compressed, (char *)g_cache.hpin.p + 64, meta.size(), hipMemcpyHostToDevice, st));
      } else {
        std::memcpy(*compressed, meta.data(), meta.size());
      }
    }
    if (hipStreamSynchronize(st) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "writing the subdomain record");
    hl_debug("compress: record written");
    if (J.owned) {
      owned_alive.erase(std::remove(owned_alive.begin(), owned_alive.end(), J.h), owned_alive.end());
      mgh_hierarchy_destroy(J.h);
    }
    byte_offset += 8 + csize;
    return MGH_SUCCESS;
  };

  // Inputs: P = nbufs - 1 subdomains are fetched ahead of the one whose record is being written.
  // At the start of iteration id the buffer of subdomain id + P is the one subdomain id - 1 used,
  // and finish(id - 1) ended with a synchronisation of its lane.
  const uint64_t P = nbufs > 0 ? (uint64_t)(nbufs - 1) : 0;
  // One subdomain arriving from the host under a REL bound: its norm is reduced piece by piece on
  // the lane's stream while the following pieces are still on the link (the norm pass is a pure
  // reduction), so the decomposition starts when the last piece has landed instead of a whole norm
  // pass later. MGH_HL_STREAM_NORM=0: norm after arrival (cross-check).
  bool first_fetched = false;
  if (dd.num == 1 && !in_dev && ebtype == MGH_REL && env_get("MGH_HL_STREAM_NORM", 1) != 0) {
    if ((rc = get_hierarchy(&pre_h, &pre_owned, dtype, dd.subdomain_shape(0), cptr, dd.subdomain_offset(0), cfg, 0)) !=
        MGH_SUCCESS)
      return cleanup(rc);
    if (pre_owned) owned_alive.push_back(pre_h);
    if (mgh_sym16_supported(pre_h)) {  // (= the fused path will run, which takes the streamed norm)
      hipStream_t lst = g_cache.lane[0].st;
      if ((rc = mgh_norm_stream_begin(pre_h, lst)) != MGH_SUCCESS) return cleanup(rc);
      const size_t bytes = total * elem, warm = (size_t)192 << 20;  // (what the level pass may still find cached)
      mgh_hierarchy *const hh = pre_h;
      const ChunkFn on_piece = [&, hh, lst](size_t off, size_t nb, hipEvent_t landed) -> int {
        HL_HIP(hipStreamWaitEvent(lst, landed, 0));
        return mgh_norm_stream_add(hh, (const char *)g_cache.in[0].p + off, nb / elem, s_d,
                                   off + nb + warm < bytes ? 1 : 0, lst);
      };
      if ((rc = copy_any(g_cache.in[0].p, original, bytes, g_cache.copy_st, &on_piece)) != MGH_SUCCESS)
        return cleanup(rc);
      if (hipEventRecord(g_cache.in_ready[0], g_cache.copy_st) != hipSuccess)
        return cleanup(hl_fail(MGH_ERR_DEVICE, "hipEventRecord"));
      first_fetched = true;
    }
  }
  for (uint64_t id = first_fetched ? 1 : 0; id < std::max<uint64_t>(P, 1) && id < dd.num; id++)
    if ((rc = fetch_sub(id)) != MGH_SUCCESS) return cleanup(rc);
  SubJob jobs[kLanes];
  if ((rc = issue(jobs[0], 0)) != MGH_SUCCESS) return cleanup(rc);
  for (uint64_t id = 0; id < dd.num; id++) {
    if (P > 0 && id + P < dd.num && (rc = fetch_sub(id + P)) != MGH_SUCCESS) return cleanup(rc);
    if (nlanes > 1 && id + 1 < dd.num && (rc = issue(jobs[(id + 1) % nlanes], id + 1)) != MGH_SUCCESS)
      return cleanup(rc);
    if ((rc = finish(jobs[id % nlanes])) != MGH_SUCCESS) return cleanup(rc);
    if (nlanes == 1 && id + 1 < dd.num && (rc = issue(jobs[0], id + 1)) != MGH_SUCCESS) return cleanup(rc);
  }
  *compressed_size = byte_offset;
  return cleanup(MGH_SUCCESS);
}

int decomposer_from_header(const fmt::Header &hd, const mgh_config &cfg, Decomposer &dd) {
  dd.D = (int)hd.shape.size();
  dd.shape = hd.shape;
  dd.decomposed = hd.dd_method != fmt::DD_NOOP;
  dd.dim = hd.dd_dim;
  dd.size = hd.dd_size;
  dd.num = 1;
  if (!dd.decomposed) return MGH_SUCCESS;
  if (dd.dim >= (uint64_t)dd.D || dd.size == 0) return hl_fail(MGH_ERR_FORMAT, "header: domain decomposition");
  if (hd.dd_method == fmt::DD_MAX_DIMENSION) {
    dd.method = MGH_DD_MAXDIM;
    dd.num = (dd.shape[dd.dim] - 1) / dd.size + 1;
  } else if (hd.dd_method == fmt::DD_BLOCK) {
    dd.method = MGH_DD_BLOCK;
    for (int d = 0; d < dd.D; d++) dd.num *= (dd.shape[d] - 1) / dd.size + 1;
  } else if (hd.dd_method == fmt::DD_VARIABLE) {
    // the header records one size only; like the reference the caller's config supplies the
    // list (DomainDecomposer.hpp:448-452)
    dd.method = MGH_DD_VARIABLE;
    if (!cfg.domain_decomposition_sizes || !cfg.num_domain_decomposition_sizes)
      return hl_fail(MGH_ERR_INVALID_ARGUMENT, "Variable domain decomposition: pass the sizes in the config");
    dd.var_sizes.assign(cfg.domain_decomposition_sizes,
                        cfg.domain_decomposition_sizes + cfg.num_domain_decomposition_sizes);
    uint64_t sum = 0;
    for (uint64_t v : dd.var_sizes) sum += v;
    if (sum != dd.shape[dd.dim]) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "Variable sizes do not add up to the extent");
    dd.num = dd.var_sizes.size();
  } else {
    return hl_fail(MGH_ERR_FORMAT, "header: unknown domain decomposition");
  }
  return MGH_SUCCESS;
}

// first bytes of a (host or device) stream on the host
int fetch_host(const void *data, size_t size, size_t want, std::vector<uint8_t> &out) {
  want = std::min(want, size);
  out.resize(want);
  if (is_device_pointer(data)) HL_TRY(dev_to_host(out.data(), data, want));
  else std::memcpy(out.data(), data, want);
  return MGH_SUCCESS;
}

int read_header(const void *data, size_t size, fmt::Header &hd, size_t &meta_size) {
  std::vector<uint8_t> pre;
  HL_TRY(fetch_host(data, size, fmt::kPreambleSize, pre));
  if (pre.size() < fmt::kPreambleSize) return hl_fail(MGH_ERR_FORMAT, "stream shorter than the preamble");
  uint64_t hs = 0;
  for (int i = 0; i < 8; i++) hs |= (uint64_t)pre[5 + i] << (8 * i);
  if (hs > size - fmt::kPreambleSize) return hl_fail(MGH_ERR_FORMAT, "header: truncated");
  std::vector<uint8_t> all;
  HL_TRY(fetch_host(data, size, fmt::kPreambleSize + hs, all));
  try {
    meta_size = fmt::parse_metadata(all.data(), all.size(), hd);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_FORMAT, e.what());
  }
  return MGH_SUCCESS;
}

template <typename T>
int decompress_impl(const fmt::Header &hd, size_t meta_size, const void *compressed, size_t csize_total,
                    void **out, const mgh_config &cfg_in, bool prealloc) {
  mgh_config cfg = cfg_in;
  HL_TRY(cache_prepare(cfg.dev_id));
  HL_TRY(trim_hierarchy_cache());
  const int dtype = hd.is_double ? MGH_DOUBLE : MGH_FLOAT;
  const size_t elem = sizeof(T);
  size_t total = 1;
  for (uint64_t e : hd.shape) total *= e;
  Decomposer dd;
  HL_TRY(decomposer_from_header(hd, cfg, dd));
  for (uint64_t id = 0; id < dd.num; id++)
    for (uint64_t e : dd.subdomain_shape(id))
      if (e < 3) return hl_fail(MGH_ERR_FORMAT, "header: subdomain with fewer than 3 nodes");
  int lossless;
  if (hd.compressor == fmt::COMP_X_HUFFMAN) lossless = MGH_LOSSLESS_HUFFMAN;
  else if (hd.compressor == fmt::COMP_X_HUFFMAN_ZSTD) lossless = MGH_LOSSLESS_HUFFMAN_ZSTD;
  else return hl_fail(MGH_ERR_FORMAT, "this lossless compressor is not supported");
  if (hd.hierarchy != fmt::HIER_MULTIDIM) return hl_fail(MGH_ERR_FORMAT, "only the multi-dimensional decomposition is supported");
  const bool in_dev = is_device_pointer(compressed);
  Pretouch pretouch;  // (joined before the first byte of the output is written, and on every way out)
  if (!prealloc) {
    if (in_dev) HL_HIP(hipMalloc(out, total * elem));
    else if (!(*out = host_alloc_large(total * elem))) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "malloc");
    // fresh host pages: faulted in beside the decoder instead of inside the device -> host copy
    if (!in_dev) pretouch.start(*out, total * elem);
  }
  // auto_pin_host_buffers (CompressionHighLevel.hpp:470-497): stream and output registered for the call
  bool in_pinned_here = false, out_pinned_here = false;
  if (cfg.auto_pin_host_buffers) {
    if (!in_dev && !is_registered_host(compressed)) {
      if (hipHostRegister(const_cast<void *>(compressed), csize_total, hipHostRegisterDefault) == hipSuccess) in_pinned_here = true;
      else (void)hipGetLastError();
    }
    if (!is_device_pointer(*out) && !is_registered_host(*out)) {
      pretouch.join();
      if (hipHostRegister(*out, total * elem, hipHostRegisterDefault) == hipSuccess) out_pinned_here = true;
      else (void)hipGetLastError();
    }
  }
  const bool pipelined = env_get("MGH_HL_PIPELINE", 1) != 0;
  const int nlanes = dd.num > 1 && pipelined ? kLanes : 1;
  mgh_hierarchy *owned_h[kLanes] = {nullptr, nullptr};  // per-subdomain hierarchy (non-uniform grid) of the lane's last job
  auto cleanup = [&](int rc) {
    for (int l = 0; l < kLanes; l++) {
      if (rc != MGH_SUCCESS || owned_h[l]) (void)hipStreamSynchronize(g_cache.lane[l].st);
      if (owned_h[l]) mgh_hierarchy_destroy(owned_h[l]);
      owned_h[l] = nullptr;
    }
    pretouch.join();
    if (in_pinned_here) (void)hipHostUnregister(const_cast<void *>(compressed));
    if (out_pinned_here) (void)hipHostUnregister(*out);
    if (rc != MGH_SUCCESS && !prealloc) {
      if (in_dev) (void)hipFree(*out); else std::free(*out);
      *out = nullptr;
    }
    return rc;
  };
  const uint64_t max_elems = dd.max_subdomain_elems();
  int rc;
  // device-resident output whose subdomains are contiguous slabs: reconstruct them in place
  const bool zero_copy = is_device_pointer(*out) && dd.all_contiguous();
  auto ensure_all = [&]() -> int {
    for (int l = 0; l < nlanes; l++) {
      Lane &L = g_cache.lane[l];
      if (!zero_copy) HL_TRY(L.sub.ensure(max_elems * elem));
      HL_TRY(L.q.ensure(max_elems * 8));
      if (hd.reorder) HL_TRY(L.q2.ensure(max_elems * 8));
    }
    return MGH_SUCCESS;
  };
  if ((rc = ensure_all()) != MGH_SUCCESS) return cleanup(rc);
  const T tol = (T)hd.tol, s = (T)hd.s, norm = (T)hd.norm;
  T local_tol = tol;
  int local_eb = hd.rel ? MGH_REL : MGH_ABS;
  if (dd.decomposed) {  // CompressionHighLevel.hpp:513-522
    local_tol = local_abs_tol<T>(local_eb, norm, tol, s, dd.num);
    local_eb = MGH_ABS;
  }
  const std::vector<std::vector<double>> *cptr = hd.uniform ? nullptr : &hd.coords;
  // The reference rebuilds the coordinates of a non-uniform grid through `(float)` when it
  // decompresses, for double data as well (CompressionHighLevel.hpp:455-462). Mirrored on request
  // only: with it an f64 non-uniform stream reconstructs exactly like stock MGARD-X does, without
  // it the coordinates the compressor used are taken at full precision.
  std::vector<std::vector<double>> coords_f32;
  if (cptr && cfg.mirror_reference_coord_cast) {
    coords_f32 = hd.coords;
    for (auto &c : coords_f32)
      for (double &x : c) x = (double)(float)x;
    cptr = &coords_f32;
  }
  // ---- subdomain pipeline (decompress_pipeline_gpu, GPUPipelines.hpp:330-520) ----
  // issue(k): everything of subdomain k up to the reconstructed dense subdomain, queued on lane
  // k % nlanes without a host synchronisation (the decoder of k + 1 runs beside the recomposition
  // of k); finish(k): the dense subdomain into its box of the output when it was not reconstructed
  // in place (for a pageable host output that copy occupies the host).
  size_t byte_offset = meta_size;
  auto issue = [&](uint64_t id) -> int {
    const int lane = (int)(id % nlanes);
    Lane &L = g_cache.lane[lane];
    hipStream_t st = L.st;
    // what the lane's previous subdomain read from the host (decode tables, record head) and its
    // hierarchy are free again once the lane has drained
    HL_HIP(hipStreamSynchronize(st));
    HL_TRY(lossless_tag_check(L.ll));
    if (owned_h[lane]) mgh_hierarchy_destroy(owned_h[lane]);
    owned_h[lane] = nullptr;
    if (csize_total - byte_offset < 8) return hl_fail(MGH_ERR_FORMAT, "subdomain record truncated");
    if (in_dev) {
      // ONE device -> host read per subdomain: the size prefix and what the lossless stage will want
      // of the record's head (chunk table for the largest subdomain, decodebook) -- every further
      // read of this subdomain is served from it (dev_to_host). A read is a queued copy plus a
      // synchronisation, ~25 us: three per subdomain were more than the decoder of a 65^3 block.
      HostPrefixState &hp = host_prefix();
      const uint8_t *at = (const uint8_t *)compressed + byte_offset;
      const bool covered = hp.base && at >= hp.base && (size_t)(at - hp.base) + 8 <= hp.bytes.size();
      if (!covered) {
        const size_t want = std::min<size_t>(csize_total - byte_offset,
                                             8 + 24 + 16 * ((max_elems - 1) / std::max<uint64_t>(hd.huff_block_size, 1) + 1) +
                                                 16 + 8 * 128 + 8 * (size_t)hd.huff_dict_size + 16);
        hp.base = nullptr;
        hp.bytes.resize(want);
        HL_TRY(aux_read(hp.bytes.data(), at, want));
        hp.base = at;
      }
    }
    std::vector<uint8_t> sz;
    HL_TRY(fetch_host((const char *)compressed + byte_offset, 8, 8, sz));
    uint64_t csize = 0;
    std::memcpy(&csize, sz.data(), 8);
    byte_offset += 8;
    if (csize > csize_total - byte_offset) return hl_fail(MGH_ERR_FORMAT, "subdomain record truncated");
    const auto sshape = dd.subdomain_shape(id);
    uint64_t n = 1;
    for (uint64_t e : sshape) n *= e;
    const char *rec = (const char *)compressed + byte_offset;
    void *sub = zero_copy ? (void *)((char *)*out + dd.linear_offset(id) * elem) : L.sub.p;
    byte_offset += csize;
    if (!((double)(n * elem) / (double)csize > 1.0)) {  // GPUPipelines.hpp:414-417
      if (csize != n * elem) return hl_fail(MGH_ERR_FORMAT, "raw subdomain record has the wrong size");
      return copy_any(sub, rec, csize, st);
    }
    const uint8_t *payload = (const uint8_t *)rec;
    uint64_t ocount = 0;
    mgh_hierarchy *h = nullptr;
    bool owned = false;
    HL_TRY(get_hierarchy(&h, &owned, dtype, sshape, cptr, dd.subdomain_offset(id), cfg, lane));
    if (owned) owned_h[lane] = h;
    // 16-bit symbols between decoder and dequantizer (a quarter of the bytes the decoder writes and
    // the two passes over the finest level read). The symbol width is chosen PER LEVEL inside
    // mgh_dequantize_recompose_sym16: the finest level reads the symbols, the levels below --
    // where the out-of-dictionary values live -- an int64 copy of the coarse corner box.
    // Subdomains below 2^25 elements keep int64: the box copy and the outlier table are three more
    // launches in a latency-bound chain (256^3: 0.86 ms with int64, 0.89 ms with symbols; 512^3: 2.16
    // vs 2.03 ms). MGH_SYM16_DECODE=0 / 1: never / always (cross-checks).
    static const long sym16_env = env_get("MGH_SYM16_DECODE", -1);
    const bool sym16_decode = sym16_env < 0 ? n >= ((uint64_t)1 << 25) : sym16_env != 0;
    bool sym16 = sym16_decode && !hd.reorder && mgh_sym16_supported(h) && hd.huff_dict_size <= 65536;
    HL_TRY(lossless_decompress(L.ll, payload, csize, lossless, (int64_t *)L.q.p, n, &ocount, st, &sym16,
                               /*sync_end=*/false));
    if (sym16)
      return mgh_dequantize_recompose_sym16(h, (const uint16_t *)L.q.p, local_eb, (double)local_tol, (double)s,
                                            (double)norm, hd.huff_dict_size, (const uint64_t *)L.ll->oidx.p,
                                            (const int64_t *)L.ll->oval.p, ocount, sub, st);
    if (hd.reorder) {
      // level-linearised integers: outliers back at their linearised positions, the
      // permutation undone, then the ordinary path with nothing left to restore
      HL_TRY(mgh_outlier_restore((int64_t *)L.q.p, n, (const uint64_t *)L.ll->oidx.p,
                                 (const int64_t *)L.ll->oval.p, ocount, st));
      HL_TRY(mgh_level_linearize(h, (const int64_t *)L.q.p, (int64_t *)L.q2.p, 1, nullptr, nullptr, 0, 0, st));
      return mgh_dequantize_recompose(h, (int64_t *)L.q2.p, local_eb, (double)local_tol, (double)s, (double)norm,
                                      hd.huff_dict_size, 1, nullptr, nullptr, 0, sub, st);
    }
    return mgh_dequantize_recompose(h, (int64_t *)L.q.p, local_eb, (double)local_tol, (double)s, (double)norm,
                                    hd.huff_dict_size, 1, (const uint64_t *)L.ll->oidx.p,
                                    (const int64_t *)L.ll->oval.p, ocount, sub, st);
  };
  auto finish = [&](uint64_t id) -> int {
    if (zero_copy) return MGH_SUCCESS;
    pretouch.join();
    Lane &L = g_cache.lane[id % nlanes];
    return copy_subdomain(dd, id, elem, L.sub.p, nullptr, *out, false, L.st);
  };
  if ((rc = issue(0)) != MGH_SUCCESS) return cleanup(rc);
  for (uint64_t id = 0; id < dd.num; id++) {
    if (nlanes > 1 && id + 1 < dd.num && (rc = issue(id + 1)) != MGH_SUCCESS) return cleanup(rc);
    if ((rc = finish(id)) != MGH_SUCCESS) return cleanup(rc);
    if (nlanes == 1 && id + 1 < dd.num && (rc = issue(id + 1)) != MGH_SUCCESS) return cleanup(rc);
  }
  for (int l = 0; l < nlanes; l++) {
    if (hipStreamSynchronize(g_cache.lane[l].st) != hipSuccess) return cleanup(hl_fail(MGH_ERR_DEVICE, "sync"));
    if ((rc = lossless_tag_check(g_cache.lane[l].ll)) != MGH_SUCCESS) return cleanup(rc);
  }
  return cleanup(MGH_SUCCESS);
}

int check_config(const mgh_config *cfg) {
  if (!cfg) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "config is NULL (use mgh_config_default)");
  if (cfg->lossless != MGH_LOSSLESS_HUFFMAN && cfg->lossless != MGH_LOSSLESS_HUFFMAN_ZSTD)
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "lossless: only Huffman and Huffman_Zstd are supported");
  if (cfg->huff_dict_size < 2 || cfg->huff_dict_size > 16384 || cfg->huff_block_size == 0)
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "huff_dict_size (2..16384) / huff_block_size");
  if (cfg->reorder != 0 && cfg->reorder != 1) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "reorder must be 0 or 1");
  int ndev = mgh_device_count();
  if (ndev <= 0) return hl_fail(MGH_ERR_NO_DEVICE, "no HIP device");
  if (cfg->dev_id < 0 || cfg->dev_id >= ndev) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "dev_id");
  return MGH_SUCCESS;
}

} // namespace

extern "C" {

void mgh_config_default(mgh_config *c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->dev_id = 0;
  c->domain_decomposition = MGH_DD_MAXDIM;
  c->domain_decomposition_dim = 0;
  c->block_size = 256;
  c->estimate_outlier_ratio = 1.0;
  c->huff_dict_size = 8192;
  c->huff_block_size = 1024 * 20;
  c->lossless = MGH_LOSSLESS_HUFFMAN;
  c->zstd_compress_level = 3;
  c->normalize_coordinates = 1;
  c->max_larget_level = std::numeric_limits<uint64_t>::max();
  c->max_memory_footprint = std::numeric_limits<uint64_t>::max();
  c->reorder = 0;
  c->mirror_reference_coord_cast = 0;
  // The reference defaults to true. Here it is opt-in: on ROCm 7.0 registering and unregistering
  // caller memory (hipHostRegister / hipHostUnregister) was observed to leave the runtime
  // treating later allocations at the same addresses as pinned -- a copy into such a buffer
  // then dies with a GPU memory fault (1 in 3 runs of the test suite). Without registration
  // pageable buffers travel through the library's own ring of pinned slots (staged_h2d / _d2h):
  // 512^3 f32 host to host 15.2 / 16.9 ms against 13.7 / 13.7 ms with registered buffers (round 6,
  // bench.py end_to_end_host) -- and registering costs 16 ms per 512 MB the first time.
  c->auto_pin_host_buffers = 0;
}

int mgh_compress(int D, int dtype, const uint64_t *shape, double tol, double s, int ebtype,
                 const void *original_data, void **compressed_data, size_t *compressed_size,
                 const void *const *coords, const mgh_config *config, int output_pre_allocated) {
  if (!shape || !original_data || !compressed_data || !compressed_size)
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (D < 1 || D > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "D must be 1..5");
  if (dtype != MGH_FLOAT && dtype != MGH_DOUBLE) return hl_fail(MGH_ERR_UNSUPPORTED_DTYPE, "dtype");
  if (ebtype != MGH_REL && ebtype != MGH_ABS) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "error_bound_type");
  if (output_pre_allocated && !*compressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "pre-allocated output is NULL");
  {
    const std::string bad = env_validate();  // MGH_* developer switches: a typo is an error
    if (!bad.empty()) return hl_fail(MGH_ERR_INVALID_ARGUMENT, bad);
  }
  HL_TRY(check_config(config));
  if (hipSetDevice(config->dev_id) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  try {
    if (dtype == MGH_FLOAT)
      return compress_impl<float>(D, dtype, shape, tol, s, ebtype, original_data, compressed_data,
                                  compressed_size, coords, *config, output_pre_allocated != 0);
    return compress_impl<double>(D, dtype, shape, tol, s, ebtype, original_data, compressed_data,
                                 compressed_size, coords, *config, output_pre_allocated != 0);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
}

static int decompress_entry(const void *compressed_data, size_t compressed_size, void **decompressed_data,
                            const mgh_config *config, int output_pre_allocated, size_t expect_bytes, int expect_dtype);

int mgh_decompress(const void *compressed_data, size_t compressed_size, void **decompressed_data,
                   const mgh_config *config, int output_pre_allocated) {
  return decompress_entry(compressed_data, compressed_size, decompressed_data, config, output_pre_allocated, 0, -1);
}

int mgh_decompress_into(const void *compressed_data, size_t compressed_size, void *out, size_t out_bytes,
                        int out_dtype, const mgh_config *config) {
  if (!out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (out_dtype != MGH_FLOAT && out_dtype != MGH_DOUBLE) return hl_fail(MGH_ERR_UNSUPPORTED_DTYPE, "dtype");
  void *dst = out;
  return decompress_entry(compressed_data, compressed_size, &dst, config, 1, out_bytes, out_dtype);
}

// expect_dtype >= 0: the caller's buffer holds expect_bytes bytes of that type -- checked against the
// header the call reads anyway, before anything is written (mgh_decompress_into)
static int decompress_entry(const void *compressed_data, size_t compressed_size, void **decompressed_data,
                            const mgh_config *config, int output_pre_allocated, size_t expect_bytes, int expect_dtype) {
  if (!compressed_data || !decompressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (output_pre_allocated && !*decompressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "pre-allocated output is NULL");
  {
    const std::string bad = env_validate();
    if (!bad.empty()) return hl_fail(MGH_ERR_INVALID_ARGUMENT, bad);
  }
  mgh_config def;
  if (!config) {
    mgh_config_default(&def);
    config = &def;
  }
  int ndev = mgh_device_count();
  if (ndev <= 0) return hl_fail(MGH_ERR_NO_DEVICE, "no HIP device");
  if (hipSetDevice(config->dev_id) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  std::unique_ptr<HostPrefix> prefix;
  if (is_device_pointer(compressed_data)) prefix.reset(new HostPrefix(compressed_data, compressed_size));
  fmt::Header hd;
  size_t meta_size = 0;
  HL_TRY(read_header(compressed_data, compressed_size, hd, meta_size));
  if (hd.shape.empty() || hd.shape.size() > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "header: dimension");
  if (!hd.quantized) return hl_fail(MGH_ERR_FORMAT, "not a compressed (quantized) stream");
  if (expect_dtype >= 0) {
    size_t need = hd.is_double ? 8 : 4;
    for (uint64_t e : hd.shape) need *= e;
    if ((expect_dtype == MGH_DOUBLE) != hd.is_double)
      return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_decompress_into: the stream holds the other data type");
    if (expect_bytes != need)
      return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_decompress_into: the buffer does not have the size of the array in the stream");
  }
  try {
    if (hd.is_double)
      return decompress_impl<double>(hd, meta_size, compressed_data, compressed_size, decompressed_data,
                                     *config, output_pre_allocated != 0);
    return decompress_impl<float>(hd, meta_size, compressed_data, compressed_size, decompressed_data,
                                  *config, output_pre_allocated != 0);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
}

}  // extern "C"

// ---- one process, several devices ------------------------------------------------------------
// The domain is cut into slabs along the slowest dimension (recorded in the header as a
// MaxDim decomposition of dimension 0, so any MGARD-X reader finds the subdomains where it
// expects them); slab id runs on device dev_ids[id % num_dev], every device on its own host
// thread with its own streams and caches. Subdomains are independent (own hierarchy, no halo:
// DomainDecomposer.hpp:260-303); the only coupling is the error budget of a REL bound: the slab
// norms are combined on the host (ErrorToleranceCalculator.hpp:69-89) and every slab then runs
// with the ABS bound calc_local_abs_tol (:134-155) -- exactly what mgh_compress does for a
// decomposed domain, so the container is the one mgh_compress would write with this decomposition
// (up to the order of the outlier lists) and mgh_decompress / mgh_decompress_multi both read it.
// Payloads are framed `[u64 size][payload]` in id order (GPUPipelines.hpp:189-193).
namespace {

struct MultiErr {
  std::mutex m;
  int rc = MGH_SUCCESS;
  std::string msg;
  void set(int code) {
    std::lock_guard<std::mutex> g(m);
    if (rc == MGH_SUCCESS) {
      rc = code;
      msg = mgh_last_error();
    }
  }
};

inline uint64_t multi_slab_size(uint64_t n0, int ndev) {
  // ceil(n0 / ndev), grown until the last slab has at least 3 planes (a hierarchy needs them)
  uint64_t size = (n0 + ndev - 1) / ndev;
  size = std::max<uint64_t>(size, 3);
  while (size < n0 && n0 % size != 0 && n0 % size < 3) size++;
  return size;
}

void worker_thread_teardown();

template <typename T>
int compress_multi_impl(int ndev, const int *devs, int D, int dtype, const uint64_t *shape,
                        double tol_d, double s_d, int ebtype, const void *original,
                        void **compressed, size_t *compressed_size, const void *const *coords_in,
                        const mgh_config &cfg0, bool prealloc) {
  const size_t elem = sizeof(T);
  Decomposer dd;
  dd.D = D;
  dd.shape.assign(shape, shape + D);
  dd.method = MGH_DD_MAXDIM;
  dd.dim = 0;
  dd.size = multi_slab_size(shape[0], ndev);
  dd.num = (shape[0] - 1) / dd.size + 1;
  dd.decomposed = dd.num > 1;
  if (!dd.decomposed) {  // nothing to share out
    mgh_config c = cfg0;
    c.dev_id = devs[0];
    return mgh_compress(D, dtype, shape, tol_d, s_d, ebtype, original, compressed, compressed_size,
                        coords_in, &c, prealloc);
  }
  size_t total = 1, inner = 1;
  for (int d = 0; d < D; d++) total *= shape[d];
  for (int d = 1; d < D; d++) inner *= shape[d];
  const T tol = (T)tol_d, s = (T)s_d;
  std::vector<std::vector<double>> coords;
  if (coords_in) {
    coords.resize(D);
    for (int d = 0; d < D; d++) {
      const T *c = static_cast<const T *>(coords_in[d]);
      coords[d].assign(c, c + shape[d]);
    }
  }
  // Device-resident input (GPUPipelines.hpp:69-207 takes device pointers): the volume lives on ONE
  // device; a slab that runs elsewhere travels there ONCE, device to device (hipMemcpyPeerAsync:
  // xGMI), into a buffer of the worker that serves both the norm and the compression; slabs of
  // the source device are compressed where they are. The container is assembled in device
  // memory of the source device (same memory space as the input, like mgh_compress).
  const bool in_dev = is_device_pointer(original);
  int src_dev = -1;
  if (in_dev) {
    hipPointerAttribute_t a;
    HL_HIP(hipPointerGetAttributes(&a, original));
    src_dev = a.device;
  }
  const uint64_t num = dd.num;
  const int nthr = (int)std::min<uint64_t>(num, (uint64_t)ndev);
  auto slab_ptr = [&](uint64_t id) { return (const char *)original + dd.linear_offset(id) * elem; };
  auto slab_coords = [&](uint64_t id, std::vector<const void *> &out) {
    out.assign(D, nullptr);
    if (!coords_in) return (const void *const *)nullptr;
    for (int d = 0; d < D; d++)
      out[d] = (const void *)(static_cast<const T *>(coords_in[d]) + (d == 0 ? dd.subdomain_offset(id)[0] : 0));
    return (const void *const *)out.data();
  };
  MultiErr err;
  // One worker thread per device for the whole call: (REL) norms of its slabs -> rendezvous, where
  // the last worker to arrive combines them on the host (ErrorToleranceCalculator.hpp:69-89) ->
  // every slab as a stand-alone ABS compression (calc_local_abs_tol).
  std::vector<double> ln(num, 0.0);
  std::vector<void *> part(num, nullptr);
  std::vector<size_t> part_size(num, 0);
  std::mutex rv_m;
  std::condition_variable rv_cv;
  int rv_arrived = 0;
  T norm = 1, local_tol = 0;
  auto rendezvous = [&] {
    std::unique_lock<std::mutex> lk(rv_m);
    if (++rv_arrived == nthr) {
      if (ebtype == MGH_REL && err.rc == MGH_SUCCESS) {
        double acc = 0;
        for (uint64_t id = 0; id < num; id++) {
          const uint64_t cnt = inner * dd.subdomain_shape(id)[0];
          if (s == std::numeric_limits<T>::infinity()) acc = std::max(acc, ln[id]);
          else acc += ln[id] * ln[id] * (cfg0.normalize_coordinates ? (double)cnt : 1.0);
        }
        if (s == std::numeric_limits<T>::infinity()) norm = (T)acc;
        else norm = (T)(cfg0.normalize_coordinates ? std::sqrt(acc / (double)total) : std::sqrt(acc));
      }
      local_tol = local_abs_tol<T>(ebtype, norm, tol, s, num);
      rv_cv.notify_all();
    } else {
      rv_cv.wait(lk, [&] { return rv_arrived == nthr; });
    }
  };
  {
    std::vector<std::thread> th;
    for (int k = 0; k < nthr; k++)
      th.emplace_back([&, k] {
        bool arrived = false;
        try {
          mgh_config c = cfg0;
          c.dev_id = devs[k];
          c.domain_decomposition = MGH_DD_MAXDIM;  // (a slab that does not fit is split further)
          bool ok = hipSetDevice(c.dev_id) == hipSuccess && cache_prepare(c.dev_id) == MGH_SUCCESS;
          if (!ok) err.set(hl_fail(MGH_ERR_DEVICE, "device of a worker thread"));
          // slabs of this worker; a device-resident slab of another device is brought over once
          // and kept when it is the worker's only one (the usual case: one slab per device)
          std::vector<uint64_t> mine;
          for (uint64_t id = k; id < num; id += nthr) mine.push_back(id);
          // the worker's copies of its remote slabs: every one travels ONCE and serves both the
          // norm and the compression; if they do not all fit the device, the oldest ones are
          // dropped and fetched again when their turn comes
          std::vector<std::pair<uint64_t, std::unique_ptr<DevBuf>>> peers;
          struct PeerGuard {  // (released on every way out of the worker, exceptions included)
            std::vector<std::pair<uint64_t, std::unique_ptr<DevBuf>>> &v;
            ~PeerGuard() {
              for (auto &kv : v) kv.second->release();
              v.clear();
            }
          } peer_guard{peers};
          auto local_ptr = [&](uint64_t id, const void **out) -> int {
            // where this worker reads slab id from: host memory, the source device itself, or a peer copy
            // (MGH_MULTI_FORCE_PEER=1: take the peer copy also when the slab is already on the
            // worker's device -- the one-GPU test of that path)
            static const bool force_peer = env_get("MGH_MULTI_FORCE_PEER", 0) != 0;
            if (!in_dev || (c.dev_id == src_dev && !force_peer)) {
              *out = slab_ptr(id);
              return MGH_SUCCESS;
            }
            for (auto &kv : peers)
              if (kv.first == id) {
                *out = kv.second->p;
                return MGH_SUCCESS;
              }
            const size_t bytes = inner * dd.subdomain_shape(id)[0] * elem;
            std::unique_ptr<DevBuf> nb(new DevBuf());
            while (nb->ensure(bytes) != MGH_SUCCESS) {
              if (peers.empty()) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "peer copy of a slab");
              peers.front().second->release();
              peers.erase(peers.begin());
            }
            HL_HIP(hipMemcpyPeerAsync(nb->p, c.dev_id, slab_ptr(id), src_dev, bytes, g_cache.lane[0].st));
            HL_HIP(hipStreamSynchronize(g_cache.lane[0].st));
            *out = nb->p;
            peers.emplace_back(id, std::move(nb));
            return MGH_SUCCESS;
          };
          if (ok && ebtype == MGH_REL) {
            for (uint64_t id : mine) {
              if (err.rc != MGH_SUCCESS) break;
              const auto sshape = dd.subdomain_shape(id);
              uint64_t cnt = 1;
              for (uint64_t e : sshape) cnt *= e;
              mgh_hierarchy *h = nullptr;
              bool owned = false;
              int rc = get_hierarchy(&h, &owned, dtype, sshape, nullptr, dd.subdomain_offset(id), c);
              const void *src = nullptr;
              if (rc == MGH_SUCCESS && in_dev) rc = local_ptr(id, &src);
              if (rc == MGH_SUCCESS && !in_dev) {
                rc = g_cache.in[0].ensure(cnt * elem);
                if (rc == MGH_SUCCESS) rc = copy_any(g_cache.in[0].p, slab_ptr(id), cnt * elem, g_cache.lane[0].st);
                src = g_cache.in[0].p;
              }
              if (rc == MGH_SUCCESS) rc = mgh_norm(h, src, s_d, &ln[id], g_cache.lane[0].st);
              if (owned) mgh_hierarchy_destroy(h);
              if (rc != MGH_SUCCESS) err.set(rc);
            }
          }
          rendezvous();
          arrived = true;
          if (ok && err.rc == MGH_SUCCESS) {
            for (uint64_t id : mine) {
              if (err.rc != MGH_SUCCESS) break;
              const auto sshape = dd.subdomain_shape(id);
              std::vector<const void *> cs;
              const void *const *cp = slab_coords(id, cs);
              const void *src = nullptr;
              int rc = local_ptr(id, &src);
              size_t sz = 0;
              if (rc == MGH_SUCCESS)
                rc = mgh_compress(D, dtype, sshape.data(), (double)local_tol, s_d, MGH_ABS, src, &part[id], &sz,
                                  cp, &c, 0);
              part_size[id] = sz;
              for (size_t k = 0; k < peers.size(); k++)  // (this slab's copy has served its purpose)
                if (peers[k].first == id) {
                  peers[k].second->release();
                  peers.erase(peers.begin() + k);
                  break;
                }
              if (rc != MGH_SUCCESS) err.set(rc);
            }
          }
          worker_thread_teardown();
        } catch (const std::exception &e) {
          // (an exception must not leave a worker thread: std::terminate)
          err.set(hl_fail(MGH_ERR_OUT_OF_MEMORY, std::string("worker thread: ") + e.what()));
          if (!arrived) rendezvous();  // the others must not wait for this thread for ever
        }
      });
    for (auto &t : th) t.join();
  }
  // (parts are host memory for a host input, device memory of their worker's device otherwise)
  auto free_parts = [&] {
    for (void *p : part) {
      if (!p) continue;
      if (in_dev) (void)hipFree(p); else std::free(p);
    }
  };
  if (err.rc != MGH_SUCCESS) {
    free_parts();
    mgh_set_last_error_(err.msg.c_str());
    return err.rc;
  }
  // ---- assembly: one header, the records in id order -------------------------------------------
  fmt::Header hdr;
  header_from(dd, dtype, ebtype, tol_d, s_d, ebtype == MGH_REL ? (double)norm : 0.0,
              coords_in ? &coords : nullptr, cfg0, hdr);
  const std::vector<uint8_t> meta = fmt::serialize_metadata(hdr);
  std::vector<size_t> body_off(num), body_len(num);
  size_t need = meta.size();
  for (uint64_t id = 0; id < num; id++) {
    fmt::Header sh;
    size_t ms = 0;
    int rc = read_header(part[id], part_size[id], sh, ms);
    if (rc == MGH_SUCCESS && sh.dd_method != fmt::DD_NOOP)
      rc = hl_fail(MGH_ERR_OUT_OF_MEMORY, "a slab does not fit its device in one piece: use more, smaller slabs");
    if (rc != MGH_SUCCESS) {
      free_parts();
      return rc;
    }
    body_off[id] = ms;
    body_len[id] = part_size[id] - ms;
    need += body_len[id];
  }
  if (in_dev && hipSetDevice(src_dev) != hipSuccess) {
    free_parts();
    return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  }
  if (!prealloc) {
    const bool ok = in_dev ? hipMalloc(compressed, need) == hipSuccess : (*compressed = std::malloc(need)) != nullptr;
    if (!ok) {
      free_parts();
      return hl_fail(MGH_ERR_OUT_OF_MEMORY, in_dev ? "hipMalloc" : "malloc");
    }
  } else if (*compressed_size < need) {
    free_parts();
    return hl_fail(MGH_ERR_OUTPUT_TOO_LARGE, "output buffer too small");
  }
  char *o = (char *)*compressed;
  const bool any_dev = in_dev || is_device_pointer(o);
  bool copied = true;
  auto put = [&](size_t at, const void *src, size_t len) {
    if (!any_dev) std::memcpy(o + at, src, len);
    else copied = copied && hipMemcpy(o + at, src, len, hipMemcpyDefault) == hipSuccess;
  };
  put(0, meta.data(), meta.size());
  size_t at = meta.size();
  for (uint64_t id = 0; id < num; id++) {
    put(at, (const char *)part[id] + body_off[id], body_len[id]);
    at += body_len[id];
  }
  free_parts();
  if (!copied) {
    if (!prealloc) {
      if (in_dev) (void)hipFree(*compressed); else std::free(*compressed);
      *compressed = nullptr;
    }
    return hl_fail(MGH_ERR_DEVICE, "assembling the container");
  }
  *compressed_size = at;
  return MGH_SUCCESS;
}

template <typename T>
int decompress_multi_impl(int ndev, const int *devs, const fmt::Header &hd, size_t meta_size,
                          const void *compressed, size_t compressed_size, void **out,
                          const mgh_config &cfg0, bool prealloc) {
  const size_t elem = sizeof(T);
  Decomposer dd;
  HL_TRY(decomposer_from_header(hd, cfg0, dd));
  size_t total = 1;
  for (uint64_t e : hd.shape) total *= e;
  if (!prealloc && !(*out = std::malloc(total * elem))) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "malloc");
  auto bail = [&](int rc) {
    if (!prealloc) {
      std::free(*out);
      *out = nullptr;
    }
    return rc;
  };
  // record offsets (the sizes are in the stream)
  const uint64_t num = dd.num;
  std::vector<size_t> off(num), len(num);
  size_t at = meta_size;
  for (uint64_t id = 0; id < num; id++) {
    if (at + 8 > compressed_size) return bail(hl_fail(MGH_ERR_FORMAT, "truncated stream"));
    uint64_t cs = 0;
    std::memcpy(&cs, (const char *)compressed + at, 8);
    if (cs > compressed_size - at - 8) return bail(hl_fail(MGH_ERR_FORMAT, "truncated record"));
    off[id] = at;
    len[id] = 8 + (size_t)cs;
    at += len[id];
  }
  const T norm = (T)hd.norm, tol = (T)hd.tol, s = (T)hd.s;
  const T local_tol = local_abs_tol<T>(hd.rel ? MGH_REL : MGH_ABS, norm, tol, s, num);
  const int nthr = (int)std::min<uint64_t>(num, (uint64_t)ndev);
  MultiErr err;
  std::vector<std::thread> th;
  for (int k = 0; k < nthr; k++)
    th.emplace_back([&, k] {
        try {
      mgh_config c = cfg0;
      c.dev_id = devs[k];
      for (uint64_t id = k; id < num && err.rc == MGH_SUCCESS; id += nthr) {
        // the record as a container of its own: header of the slab (ABS bound) + the record
        Decomposer one;
        one.D = dd.D;
        one.shape = dd.subdomain_shape(id);
        one.decomposed = false;
        one.num = 1;
        std::vector<std::vector<double>> sc;
        if (!hd.uniform) {
          sc = hd.coords;
          const uint64_t o0 = dd.subdomain_offset(id)[0];
          sc[0].assign(hd.coords[0].begin() + o0, hd.coords[0].begin() + o0 + one.shape[0]);
        }
        // everything the stream says about itself (reorder, lossless choice, dictionary, ...) comes
        // from ITS header, never from the caller's config; only what describes the slab changes
        fmt::Header sh = hd;
        sh.shape = one.shape;
        if (!hd.uniform) sh.coords = sc;
        sh.rel = false;
        sh.tol = (double)local_tol;
        sh.norm = 0.0;
        sh.dd_method = fmt::DD_NOOP;
        sh.dd_dim = 0;
        sh.dd_size = 0;
        const std::vector<uint8_t> meta = fmt::serialize_metadata(sh);
        std::vector<uint8_t> mini(meta.size() + len[id]);
        std::memcpy(mini.data(), meta.data(), meta.size());
        std::memcpy(mini.data() + meta.size(), (const char *)compressed + off[id], len[id]);
        void *dst = (char *)*out + dd.linear_offset(id) * elem;
        const int rc = mgh_decompress(mini.data(), mini.size(), &dst, &c, 1);
        if (rc != MGH_SUCCESS) err.set(rc);
      }
      worker_thread_teardown();
      } catch (const std::exception &e) {
          // (an exception must not leave a worker thread: std::terminate)
          err.set(hl_fail(MGH_ERR_OUT_OF_MEMORY, std::string("worker thread: ") + e.what()));
        }
      });
  for (auto &t : th) t.join();
  if (err.rc != MGH_SUCCESS) {
    mgh_set_last_error_(err.msg.c_str());
    return bail(err.rc);
  }
  return MGH_SUCCESS;
}

// End of a worker thread of the multi-device calls: release the device state AND delete the
// per-thread objects themselves (they are deliberately not destroyed by thread_local destructors --
// HIP calls at process exit -- so a thread that ends has to do it, or every call leaks them).
void worker_thread_teardown() {
  mgh_release_cache();
  delete g_cache_ptr;
  g_cache_ptr = nullptr;
}

int check_devs(int num_dev, const int *dev_ids) {
  if (num_dev < 1 || !dev_ids) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "device list");
  const int have = mgh_device_count();
  if (have <= 0) return hl_fail(MGH_ERR_NO_DEVICE, "no HIP device");
  for (int k = 0; k < num_dev; k++)
    if (dev_ids[k] < 0 || dev_ids[k] >= have) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "device id out of range");
  return MGH_SUCCESS;
}

}  // namespace

extern "C" {

int mgh_compress_multi(int num_dev, const int *dev_ids, int D, int dtype, const uint64_t *shape,
                       double tol, double s, int ebtype, const void *original_data,
                       void **compressed_data, size_t *compressed_size, const void *const *coords,
                       const mgh_config *config, int output_pre_allocated) {
  if (!shape || !original_data || !compressed_data || !compressed_size)
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (D < 1 || D > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "D must be 1..5");
  if (dtype != MGH_FLOAT && dtype != MGH_DOUBLE) return hl_fail(MGH_ERR_UNSUPPORTED_DTYPE, "dtype");
  if (ebtype != MGH_REL && ebtype != MGH_ABS) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "error_bound_type");
  if (output_pre_allocated && !*compressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "pre-allocated output is NULL");
  HL_TRY(check_devs(num_dev, dev_ids));
  HL_TRY(check_config(config));
  // host input -> host container; device-resident input (on any one device) -> container in device
  // memory of that device. A pre-allocated output must be of the same kind.
  if (output_pre_allocated && is_device_pointer(original_data) != is_device_pointer(*compressed_data))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_compress_multi: input and pre-allocated output must both be host or both be device memory");
  try {
    if (dtype == MGH_FLOAT)
      return compress_multi_impl<float>(num_dev, dev_ids, D, dtype, shape, tol, s, ebtype, original_data,
                                        compressed_data, compressed_size, coords, *config,
                                        output_pre_allocated != 0);
    return compress_multi_impl<double>(num_dev, dev_ids, D, dtype, shape, tol, s, ebtype, original_data,
                                       compressed_data, compressed_size, coords, *config,
                                       output_pre_allocated != 0);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
}

int mgh_decompress_multi(int num_dev, const int *dev_ids, const void *compressed_data,
                         size_t compressed_size, void **decompressed_data, const mgh_config *config,
                         int output_pre_allocated) {
  if (!compressed_data || !decompressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (output_pre_allocated && !*decompressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "pre-allocated output is NULL");
  HL_TRY(check_devs(num_dev, dev_ids));
  mgh_config def;
  if (!config) {
    mgh_config_default(&def);
    config = &def;
  }
  if (is_device_pointer(compressed_data) || (output_pre_allocated && is_device_pointer(*decompressed_data)))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_decompress_multi: host buffers only");
  fmt::Header hd;
  size_t meta_size = 0;
  HL_TRY(read_header(compressed_data, compressed_size, hd, meta_size));
  if (hd.shape.empty() || hd.shape.size() > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "header: dimension");
  if (!hd.quantized) return hl_fail(MGH_ERR_FORMAT, "not a compressed (quantized) stream");
  // slabs of the slowest dimension only; anything else goes through the single-device path
  const bool slabs = hd.dd_method == fmt::DD_MAX_DIMENSION && hd.dd_dim == 0;
  if (!slabs) {
    mgh_config c = *config;
    c.dev_id = dev_ids[0];
    return mgh_decompress(compressed_data, compressed_size, decompressed_data, &c, output_pre_allocated);
  }
  try {
    if (hd.is_double)
      return decompress_multi_impl<double>(num_dev, dev_ids, hd, meta_size, compressed_data, compressed_size,
                                           decompressed_data, *config, output_pre_allocated != 0);
    return decompress_multi_impl<float>(num_dev, dev_ids, hd, meta_size, compressed_data, compressed_size,
                                        decompressed_data, *config, output_pre_allocated != 0);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
}

}  // extern "C"

// ---- one rank per GPU: RCCL ---------------------------------------------------------------------
// The reference's multi-GPU pattern is one MPI rank per GPU, every rank compressing its own block
// (examples/mgard-x/CompressXgcData/TestXGCAbsoluteError.cpp:36-252). Here the ranks' slabs of the
// slowest dimension form ONE domain: the norm a REL bound refers to is reduced over the ranks
// (ncclAllReduce: MAX for s = inf, SUM of squares otherwise -- ErrorToleranceCalculator.hpp:69-131),
// every rank compresses its slab with the ABS bound of calc_local_abs_tol (:134-155), the record
// sizes travel by ncclAllGather and the records by ncclSend / ncclRecv to the root, which writes the
// container mgh_compress would write for this decomposition (GPUPipelines.hpp:189-193). RCCL is
// resolved at first use (dlsym of an RCCL the application already carries, else librccl.so.1), so
// single-GPU users do not link it; the communicator is the caller's.
namespace {
struct RcclApi {
  void *lib = nullptr;
  bool ready = false;
  // (ncclResult_t, ncclDataType_t, ncclRedOp_t are ints; ncclComm_t is a pointer: rccl.h:448-470)
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  static constexpr int kUint8 = 1, kUint64 = 5, kFloat64 = 8, kSum = 0, kMax = 2;
  bool bind(void *from) {
    auto sym = [&](const char *n) { return dlsym(from, n); };
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    ready = AllReduce && AllGather && Broadcast && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    return ready;
  }
  bool load(const char *path) {
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    if (path) {  // the library the caller's communicator comes from
      void *l = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
      if (!l) return false;
      lib = l;
      return bind(l);
    }
    if (ready) return true;
    if (bind(RTLD_DEFAULT)) return true;  // the application links RCCL itself
    for (const char *n : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (lib && bind(lib)) return true;
    }
    return false;
  }
};
RcclApi g_rccl;

#define HL_NCCL(expr)                                                                            \
  do {                                                                                           \
    const int _r = (expr);                                                                       \
    if (_r != 0) return hl_fail(MGH_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); \
  } while (0)

// slabs of dimension 0 in rank order as a MaxDim decomposition: all of one size, the last one
// may be shorter
int dist_slab_size(const std::vector<uint64_t> &n0, uint64_t *size) {
  *size = n0[0];
  for (size_t r = 0; r + 1 < n0.size(); r++)
    if (n0[r] != *size) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_*_dist: every rank but the last must hold the same number of planes");
  if (n0.back() > *size || n0.back() < 3) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_*_dist: the last rank holds more planes than the others, or fewer than 3");
  return MGH_SUCCESS;
}

template <typename T>
int compress_dist_impl(void *comm, int rank, int nranks, int root, int D, int dtype, const uint64_t *lshape,
                       double tol_d, double s_d, int ebtype, const void *d_local, void **compressed,
                       size_t *compressed_size, const void *const *coords_in, const mgh_config &cfg, bool prealloc) {
  HL_TRY(cache_prepare(cfg.dev_id));
  hipStream_t st = g_cache.lane[0].st;
  const T tol = (T)tol_d, s = (T)s_d;
  // ---- every rank learns every slab's shape (and that they agree on everything else) ----
  constexpr int kMeta = 8;
  DevBuf dmeta;
  struct Rel { DevBuf &b; ~Rel() { b.release(); } } rel_meta{dmeta};
  HL_TRY(dmeta.ensure((size_t)(nranks + 1) * kMeta * 8));
  uint64_t mine[kMeta] = {(uint64_t)D, (uint64_t)dtype, 0, 0, 0, 0, 0, coords_in ? 1u : 0u};
  for (int d = 0; d < D; d++) mine[2 + d] = lshape[d];
  uint64_t *d_mine = (uint64_t *)dmeta.p, *d_all = d_mine + kMeta;
  HL_HIP(hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, st));
  HL_NCCL(g_rccl.AllGather(d_mine, d_all, kMeta, RcclApi::kUint64, comm, st));
  std::vector<uint64_t> all((size_t)nranks * kMeta);
  HL_HIP(hipMemcpyAsync(all.data(), d_all, all.size() * 8, hipMemcpyDeviceToHost, st));
  HL_HIP(hipStreamSynchronize(st));
  std::vector<uint64_t> n0(nranks);
  for (int r = 0; r < nranks; r++) {
    const uint64_t *m = &all[(size_t)r * kMeta];
    bool same = m[0] == mine[0] && m[1] == mine[1] && m[7] == mine[7];
    for (int d = 1; d < D; d++) same = same && m[2 + d] == mine[2 + d];
    if (!same) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_compress_dist: the ranks disagree on dimension, type or the extents of dimensions 1..");
    n0[r] = m[2];
  }
  if (nranks == 1)  // (nothing to share out: the plain call, like mgh_compress_multi on one slab)
    return mgh_compress(D, dtype, lshape, tol_d, s_d, ebtype, d_local, compressed, compressed_size, coords_in, &cfg,
                        prealloc);
  uint64_t slab = 0;
  HL_TRY(dist_slab_size(n0, &slab));
  Decomposer dd;
  dd.D = D;
  dd.shape.assign(lshape, lshape + D);
  dd.shape[0] = 0;
  for (uint64_t e : n0) dd.shape[0] += e;
  dd.method = MGH_DD_MAXDIM;
  dd.dim = 0;
  dd.size = slab;
  dd.num = (uint64_t)nranks;
  dd.decomposed = true;
  size_t total = 1, cnt = 1;
  for (int d = 0; d < D; d++) {
    total *= dd.shape[d];
    cnt *= lshape[d];
  }
  // ---- the norm of the whole domain ----
  T norm = 1;
  DevBuf dscal;
  Rel rel_scal{dscal};
  HL_TRY(dscal.ensure(64));
  if (ebtype == MGH_REL) {
    mgh_hierarchy *h = nullptr;
    bool owned = false;
    std::vector<uint64_t> sshape(lshape, lshape + D), off(D, 0);
    HL_TRY(get_hierarchy(&h, &owned, dtype, sshape, nullptr, off, cfg, 0));
    double ln = 0;
    const int rc = mgh_norm(h, d_local, s_d, &ln, st);
    if (owned) mgh_hierarchy_destroy(h);
    HL_TRY(rc);
    const bool inf = s == std::numeric_limits<T>::infinity();
    double acc = inf ? ln : ln * ln * (cfg.normalize_coordinates ? (double)cnt : 1.0);
    HL_HIP(hipMemcpyAsync(dscal.p, &acc, 8, hipMemcpyHostToDevice, st));
    HL_NCCL(g_rccl.AllReduce(dscal.p, (char *)dscal.p + 8, 1, RcclApi::kFloat64, inf ? RcclApi::kMax : RcclApi::kSum, comm, st));
    HL_HIP(hipMemcpyAsync(&acc, (char *)dscal.p + 8, 8, hipMemcpyDeviceToHost, st));
    HL_HIP(hipStreamSynchronize(st));
    if (inf) norm = (T)acc;
    else norm = (T)(cfg.normalize_coordinates ? std::sqrt(acc / (double)total) : std::sqrt(acc));
  }
  const T local_tol = local_abs_tol<T>(ebtype, norm, tol, s, (uint64_t)nranks);
  // ---- this rank's slab as a stand-alone ABS compression; its body is the record ----
  void *part = nullptr;
  size_t part_size = 0;
  mgh_config c = cfg;
  c.domain_decomposition = MGH_DD_MAXDIM;
  HL_TRY(mgh_compress(D, dtype, lshape, (double)local_tol, s_d, MGH_ABS, d_local, &part, &part_size, coords_in, &c, 0));
  struct FreePart { void *p; ~FreePart() { if (p) (void)hipFree(p); } } free_part{part};
  fmt::Header sh;
  size_t ms = 0;
  HL_TRY(read_header(part, part_size, sh, ms));
  if (sh.dd_method != fmt::DD_NOOP) return hl_fail(MGH_ERR_OUT_OF_MEMORY, "mgh_compress_dist: the slab does not fit its device in one piece");
  uint64_t body = part_size - ms;
  HL_HIP(hipMemcpyAsync(d_mine, &body, 8, hipMemcpyHostToDevice, st));
  HL_NCCL(g_rccl.AllGather(d_mine, d_all, 1, RcclApi::kUint64, comm, st));
  std::vector<uint64_t> len(nranks);
  HL_HIP(hipMemcpyAsync(len.data(), d_all, (size_t)nranks * 8, hipMemcpyDeviceToHost, st));
  // coordinates of dimension 0: every rank's slab of them to the root (padded to the slab size)
  std::vector<double> c0;
  if (coords_in) {
    DevBuf dc;
    Rel rel_c{dc};
    HL_TRY(dc.ensure((size_t)(nranks + 1) * slab * 8));
    std::vector<double> my(slab, 0.0);
    for (uint64_t i = 0; i < lshape[0]; i++) my[i] = (double)static_cast<const T *>(coords_in[0])[i];
    HL_HIP(hipMemcpyAsync(dc.p, my.data(), slab * 8, hipMemcpyHostToDevice, st));
    HL_NCCL(g_rccl.AllGather(dc.p, (char *)dc.p + slab * 8, slab, RcclApi::kFloat64, comm, st));
    std::vector<double> allc((size_t)nranks * slab);
    HL_HIP(hipMemcpyAsync(allc.data(), (char *)dc.p + slab * 8, allc.size() * 8, hipMemcpyDeviceToHost, st));
    HL_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < nranks; r++) c0.insert(c0.end(), allc.begin() + (size_t)r * slab, allc.begin() + (size_t)r * slab + n0[r]);
  }
  HL_HIP(hipStreamSynchronize(st));
  if (rank != root) {
    HL_NCCL(g_rccl.Send((const char *)part + ms, body, RcclApi::kUint8, root, comm, st));
    HL_HIP(hipStreamSynchronize(st));
    *compressed_size = 0;
    return MGH_SUCCESS;
  }
  // ---- root: header of the whole domain, then the records in rank order ----
  std::vector<std::vector<double>> coords;
  if (coords_in) {
    coords.resize(D);
    coords[0] = c0;
    for (int d = 1; d < D; d++) {
      const T *cd = static_cast<const T *>(coords_in[d]);
      coords[d].assign(cd, cd + lshape[d]);
    }
  }
  fmt::Header hdr;
  header_from(dd, dtype, ebtype, tol_d, s_d, ebtype == MGH_REL ? (double)norm : 0.0, coords_in ? &coords : nullptr, cfg, hdr);
  const std::vector<uint8_t> meta = fmt::serialize_metadata(hdr);
  size_t need = meta.size();
  for (uint64_t l : len) need += l;
  if (!prealloc) HL_HIP(hipMalloc(compressed, need));
  else if (*compressed_size < need) return hl_fail(MGH_ERR_OUTPUT_TOO_LARGE, "output buffer too small");
  else if (!is_device_pointer(*compressed)) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_compress_dist: the container is assembled in device memory");
  char *o = (char *)*compressed;
  auto bail = [&](int rc) {
    if (!prealloc) {
      (void)hipFree(*compressed);
      *compressed = nullptr;
    }
    return rc;
  };
  HL_TRY(g_cache.hpin.ensure(64 + meta.size()));
  std::memcpy((char *)g_cache.hpin.p + 64, meta.data(), meta.size());
  if (hipMemcpyAsync(o, (char *)g_cache.hpin.p + 64, meta.size(), hipMemcpyHostToDevice, st) != hipSuccess)
    return bail(hl_fail(MGH_ERR_DEVICE, "header"));
  size_t at = meta.size();
  int nrc = g_rccl.GroupStart();
  for (int r = 0; r < nranks && nrc == 0; r++) {
    if (r != root) nrc = g_rccl.Recv(o + at, len[r], RcclApi::kUint8, r, comm, st);
    at += len[r];
  }
  if (nrc == 0) nrc = g_rccl.GroupEnd(); else (void)g_rccl.GroupEnd();
  if (nrc != 0) return bail(hl_fail(MGH_ERR_DEVICE, std::string("ncclRecv: ") + g_rccl.GetErrorString(nrc)));
  size_t mine_at = meta.size();
  for (int r = 0; r < root; r++) mine_at += len[r];
  if (hipMemcpyAsync(o + mine_at, (const char *)part + ms, body, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return bail(hl_fail(MGH_ERR_DEVICE, "assembling the container"));
  *compressed_size = at;
  return MGH_SUCCESS;
}

template <typename T>
int decompress_dist_impl(void *comm, int rank, int nranks, int root, fmt::Header &hd, const void *compressed,
                         size_t compressed_size, size_t meta_size, void *d_local_out, const mgh_config &cfg) {
  hipStream_t st = g_cache.lane[0].st;
  Decomposer dd;
  HL_TRY(decomposer_from_header(hd, cfg, dd));
  if (!(hd.dd_method == fmt::DD_MAX_DIMENSION && hd.dd_dim == 0 && dd.num == (uint64_t)nranks))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_decompress_dist: the container is not one slab of dimension 0 per rank");
  // record table from the root
  DevBuf dtab;
  struct Rel { DevBuf &b; ~Rel() { b.release(); } } rel_tab{dtab};
  HL_TRY(dtab.ensure((size_t)nranks * 16));
  std::vector<uint64_t> tab((size_t)nranks * 2, 0);  // offset (of the size prefix), length (prefix included)
  if (rank == root) {
    size_t at = meta_size;
    for (int r = 0; r < nranks; r++) {
      if (at + 8 > compressed_size) return hl_fail(MGH_ERR_FORMAT, "truncated stream");
      uint64_t cs = 0;
      HL_TRY(aux_read(&cs, (const char *)compressed + at, 8));
      if (cs > compressed_size - at - 8) return hl_fail(MGH_ERR_FORMAT, "truncated record");
      tab[2 * r] = at;
      tab[2 * r + 1] = 8 + cs;
      at += 8 + cs;
    }
    HL_HIP(hipMemcpyAsync(dtab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));
  }
  HL_NCCL(g_rccl.Broadcast(dtab.p, dtab.p, tab.size(), RcclApi::kUint64, root, comm, st));
  HL_HIP(hipMemcpyAsync(tab.data(), dtab.p, tab.size() * 8, hipMemcpyDeviceToHost, st));
  HL_HIP(hipStreamSynchronize(st));
  // this rank's record behind the header of its slab: a container of its own (decompress_multi_impl)
  const T norm = (T)hd.norm, tol = (T)hd.tol, s = (T)hd.s;
  const T local_tol = local_abs_tol<T>(hd.rel ? MGH_REL : MGH_ABS, norm, tol, s, (uint64_t)nranks);
  fmt::Header sh = hd;
  sh.shape = dd.subdomain_shape((uint64_t)rank);
  if (!hd.uniform) {
    const uint64_t o0 = dd.subdomain_offset((uint64_t)rank)[0];
    sh.coords[0].assign(hd.coords[0].begin() + o0, hd.coords[0].begin() + o0 + sh.shape[0]);
  }
  sh.rel = false;
  sh.tol = (double)local_tol;
  sh.norm = 0.0;
  sh.dd_method = fmt::DD_NOOP;
  sh.dd_dim = 0;
  sh.dd_size = 0;
  const std::vector<uint8_t> meta = fmt::serialize_metadata(sh);
  const size_t mylen = tab[2 * rank + 1];
  DevBuf mini;
  Rel rel_mini{mini};
  HL_TRY(mini.ensure(meta.size() + mylen));
  HL_TRY(g_cache.hpin.ensure(64 + meta.size()));
  std::memcpy((char *)g_cache.hpin.p + 64, meta.data(), meta.size());
  HL_HIP(hipMemcpyAsync(mini.p, (char *)g_cache.hpin.p + 64, meta.size(), hipMemcpyHostToDevice, st));
  if (rank == root) {
    HL_NCCL(g_rccl.GroupStart());
    int nrc = 0;
    for (int r = 0; r < nranks && nrc == 0; r++)
      if (r != root) nrc = g_rccl.Send((const char *)compressed + tab[2 * r], tab[2 * r + 1], RcclApi::kUint8, r, comm, st);
    const int erc = g_rccl.GroupEnd();
    if (nrc != 0 || erc != 0) return hl_fail(MGH_ERR_DEVICE, std::string("ncclSend: ") + g_rccl.GetErrorString(nrc ? nrc : erc));
    HL_HIP(hipMemcpyAsync((char *)mini.p + meta.size(), (const char *)compressed + tab[2 * rank], mylen, hipMemcpyDeviceToDevice, st));
  } else {
    HL_NCCL(g_rccl.Recv((char *)mini.p + meta.size(), mylen, RcclApi::kUint8, root, comm, st));
  }
  HL_HIP(hipStreamSynchronize(st));
  void *dst = d_local_out;
  return mgh_decompress(mini.p, meta.size() + mylen, &dst, &cfg, 1);
}
}  // namespace

extern "C" {

int mgh_dist_use_library(const char *path) {
  if (!path) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (!g_rccl.load(path)) return hl_fail(MGH_ERR_INVALID_ARGUMENT, std::string("mgh_dist_use_library: no RCCL in ") + path);
  return MGH_SUCCESS;
}

int mgh_compress_dist(void *nccl_comm, int rank, int nranks, int root, int D, int dtype, const uint64_t *local_shape,
                      double tol, double s, int ebtype, const void *d_local_data, void **compressed_data,
                      size_t *compressed_size, const void *const *coords, const mgh_config *config,
                      int output_pre_allocated) {
  if (!nccl_comm || !local_shape || !d_local_data || !compressed_size) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "rank / nranks / root");
  if (rank == root && (!compressed_data || (output_pre_allocated && !*compressed_data)))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "the root needs an output");
  if (D < 1 || D > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "D must be 1..5");
  if (dtype != MGH_FLOAT && dtype != MGH_DOUBLE) return hl_fail(MGH_ERR_UNSUPPORTED_DTYPE, "dtype");
  if (ebtype != MGH_REL && ebtype != MGH_ABS) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "error_bound_type");
  HL_TRY(check_config(config));
  if (hipSetDevice(config->dev_id) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  if (!is_device_pointer_on(d_local_data, config->dev_id))
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_compress_dist: the slab must be resident on config->dev_id");
  if (!g_rccl.load(nullptr)) return hl_fail(MGH_ERR_NO_DEVICE, "mgh_compress_dist: no RCCL (librccl.so.1) found");
  size_t dummy = 0;
  void *none = nullptr;
  try {
    if (dtype == MGH_FLOAT)
      return compress_dist_impl<float>(nccl_comm, rank, nranks, root, D, dtype, local_shape, tol, s, ebtype, d_local_data,
                                       rank == root ? compressed_data : &none, rank == root ? compressed_size : &dummy,
                                       coords, *config, rank == root && output_pre_allocated != 0);
    return compress_dist_impl<double>(nccl_comm, rank, nranks, root, D, dtype, local_shape, tol, s, ebtype, d_local_data,
                                      rank == root ? compressed_data : &none, rank == root ? compressed_size : &dummy,
                                      coords, *config, rank == root && output_pre_allocated != 0);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
}

int mgh_decompress_dist(void *nccl_comm, int rank, int nranks, int root, const void *compressed_data,
                        size_t compressed_size, void *d_local_out, const mgh_config *config) {
  if (!nccl_comm || !d_local_out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "rank / nranks / root");
  if (rank == root && !compressed_data) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "the root needs the container");
  mgh_config def;
  if (!config) {
    mgh_config_default(&def);
    config = &def;
  }
  if (hipSetDevice(config->dev_id) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  if (!g_rccl.load(nullptr)) return hl_fail(MGH_ERR_NO_DEVICE, "mgh_decompress_dist: no RCCL (librccl.so.1) found");
  HL_TRY(cache_prepare(config->dev_id));
  hipStream_t st = g_cache.lane[0].st;
  try {
    // the header travels first: its length, then its bytes
    if (nranks > 1 && rank == root && !is_device_pointer_on(compressed_data, config->dev_id))
      return hl_fail(MGH_ERR_INVALID_ARGUMENT, "mgh_decompress_dist: the container must be resident on the root's device");
    fmt::Header hd;
    size_t meta_size = 0;
    std::vector<uint8_t> hbytes;
    if (rank == root) {
      std::unique_ptr<HostPrefix> prefix;
      if (is_device_pointer(compressed_data)) prefix.reset(new HostPrefix(compressed_data, compressed_size));
      HL_TRY(read_header(compressed_data, compressed_size, hd, meta_size));
      HL_TRY(fetch_host(compressed_data, compressed_size, meta_size, hbytes));
    }
    DevBuf dh;
    struct Rel { DevBuf &b; ~Rel() { b.release(); } } rel{dh};
    HL_TRY(dh.ensure(8));
    uint64_t ms64 = meta_size;
    HL_HIP(hipMemcpyAsync(dh.p, &ms64, 8, hipMemcpyHostToDevice, st));
    HL_NCCL(g_rccl.Broadcast(dh.p, dh.p, 1, RcclApi::kUint64, root, nccl_comm, st));
    HL_HIP(hipMemcpyAsync(&ms64, dh.p, 8, hipMemcpyDeviceToHost, st));
    HL_HIP(hipStreamSynchronize(st));
    if (ms64 == 0 || ms64 > ((uint64_t)1 << 32)) return hl_fail(MGH_ERR_FORMAT, "header size");
    HL_TRY(dh.ensure(ms64));
    hbytes.resize(ms64);
    if (rank == root) HL_HIP(hipMemcpyAsync(dh.p, hbytes.data(), ms64, hipMemcpyHostToDevice, st));
    HL_NCCL(g_rccl.Broadcast(dh.p, dh.p, ms64, RcclApi::kUint8, root, nccl_comm, st));
    HL_HIP(hipMemcpyAsync(hbytes.data(), dh.p, ms64, hipMemcpyDeviceToHost, st));
    HL_HIP(hipStreamSynchronize(st));
    if (rank != root) {
      try {
        meta_size = fmt::parse_metadata(hbytes.data(), hbytes.size(), hd);
      } catch (const std::exception &e) {
        return hl_fail(MGH_ERR_FORMAT, e.what());
      }
    }
    if (nranks == 1) {  // (nothing to hand out: the plain call)
      void *dst = d_local_out;
      return mgh_decompress(compressed_data, compressed_size, &dst, config, 1);
    }
    if (hd.is_double)
      return decompress_dist_impl<double>(nccl_comm, rank, nranks, root, hd, compressed_data, compressed_size, meta_size,
                                          d_local_out, *config);
    return decompress_dist_impl<float>(nccl_comm, rank, nranks, root, hd, compressed_data, compressed_size, meta_size,
                                       d_local_out, *config);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
}

}  // extern "C"

extern "C" {

int mgh_infer_shape(const void *data, size_t size, int *D_out, uint64_t *shape_out) {
  if (!data || !D_out || !shape_out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  fmt::Header hd;
  size_t ms = 0;
  HL_TRY(read_header(data, size, hd, ms));
  if (hd.shape.size() > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "header: dimension");
  *D_out = (int)hd.shape.size();
  for (size_t d = 0; d < hd.shape.size(); d++) shape_out[d] = hd.shape[d];
  return MGH_SUCCESS;
}

int mgh_infer_data_type(const void *data, size_t size, int *dtype_out) {
  if (!data || !dtype_out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  fmt::Header hd;
  size_t ms = 0;
  HL_TRY(read_header(data, size, hd, ms));
  *dtype_out = hd.is_double ? MGH_DOUBLE : MGH_FLOAT;
  return MGH_SUCCESS;
}

// pin_memory / check_memory_pinned / unpin_memory (compress_x.hpp:166-178; HIP backend:
// MemoryManager<HIP>::HostRegister / CheckHostRegister / HostUnregister, DeviceAdapterHip.h)
int mgh_pin_memory(void *ptr, size_t num_bytes) {
  if (!ptr || !num_bytes) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (is_registered_host(ptr)) return MGH_SUCCESS;  // already pinned
  HL_HIP(hipHostRegister(ptr, num_bytes, hipHostRegisterPortable));
  return MGH_SUCCESS;
}

int mgh_check_memory_pinned(const void *ptr) { return ptr && is_registered_host(ptr) ? 1 : 0; }

int mgh_unpin_memory(void *ptr) {
  if (!ptr) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  HL_HIP(hipHostUnregister(ptr));
  return MGH_SUCCESS;
}

void mgh_free_device(void *p) {
  if (p) (void)hipFree(p);
}

void mgh_release_cache(void) {
  if (g_cache_ptr) g_cache_ptr->release();
  release_host_transfer_state();  // pinned rings, copy threads
}

int mgh_memcpy(void *dst, const void *src, size_t bytes) {
  if (!dst || !src) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  HL_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
  return MGH_SUCCESS;
}

int64_t mgh_metadata_serialize(const mgh_header_info *in, uint8_t *out, uint64_t capacity) {
  if (!in || in->D < 1 || in->D > MGH_MAX_DIM) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "header info");
  fmt::Header h;
  for (int i = 0; i < 3; i++) h.version[i] = in->version[i];
  h.is_double = in->dtype == MGH_DOUBLE;
  h.shape.assign(in->shape, in->shape + in->D);
  h.uniform = in->uniform != 0;
  if (!h.uniform)
    for (int d = 0; d < in->D; d++) {
      if (!in->coords[d]) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "coords");
      h.coords.emplace_back(in->coords[d], in->coords[d] + in->shape[d]);
    }
  h.rel = in->error_bound_type == MGH_REL;
  h.tol = in->tol;
  h.s = in->s;
  h.norm = in->norm;
  h.dd_method = !in->domain_decomposed ? fmt::DD_NOOP
                : in->dd_method == MGH_DD_MAXDIM ? fmt::DD_MAX_DIMENSION
                : in->dd_method == MGH_DD_BLOCK ? fmt::DD_BLOCK : fmt::DD_VARIABLE;
  h.dd_dim = in->dd_dim;
  h.dd_size = in->dd_size;
  h.l_target = in->l_target;
  h.reorder = in->reorder != 0;
  h.compressor = in->lossless == MGH_LOSSLESS_HUFFMAN ? fmt::COMP_X_HUFFMAN
                 : in->lossless == MGH_LOSSLESS_HUFFMAN_LZ4 ? fmt::COMP_X_HUFFMAN_LZ4
                 : in->lossless == MGH_LOSSLESS_HUFFMAN_ZSTD ? fmt::COMP_X_HUFFMAN_ZSTD
                 : fmt::COMP_CPU_HUFFMAN_ZSTD;
  h.huff_dict_size = in->lossless == MGH_LOSSLESS_CPU ? 0 : in->huff_dict_size;
  h.huff_block_size = in->lossless == MGH_LOSSLESS_CPU ? 0 : in->huff_block_size;
  const std::vector<uint8_t> b = fmt::serialize_metadata(h);
  if (out) {
    if (capacity < b.size()) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "capacity");
    std::memcpy(out, b.data(), b.size());
  }
  return (int64_t)b.size();
}

int mgh_metadata_parse(const uint8_t *data, uint64_t size, mgh_header_info *out, double *cstore,
                       uint64_t ccap, uint64_t *metadata_size_out) {
  if (!data || !out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  fmt::Header h;
  size_t ms = 0;
  try {
    ms = fmt::parse_metadata(data, size, h);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_FORMAT, e.what());
  }
  if (h.shape.empty() || h.shape.size() > MGH_MAX_DIM) return hl_fail(MGH_ERR_UNSUPPORTED_DIMENSION, "header: dimension");
  std::memset(out, 0, sizeof(*out));
  for (int i = 0; i < 3; i++) out->version[i] = h.version[i];
  out->dtype = h.is_double ? MGH_DOUBLE : MGH_FLOAT;
  out->D = (int)h.shape.size();
  for (int d = 0; d < out->D; d++) out->shape[d] = h.shape[d];
  out->uniform = h.uniform ? 1 : 0;
  if (!h.uniform) {
    uint64_t need = 0;
    for (uint64_t n : h.shape) need += n;
    if (!cstore || ccap < need) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "coords_storage too small");
    uint64_t off = 0;
    for (int d = 0; d < out->D; d++) {
      std::memcpy(cstore + off, h.coords[d].data(), h.coords[d].size() * 8);
      out->coords[d] = cstore + off;
      off += h.shape[d];
    }
  }
  out->error_bound_type = h.rel ? MGH_REL : MGH_ABS;
  out->tol = h.tol;
  out->s = h.s;
  out->norm = h.norm;
  out->domain_decomposed = h.dd_method != fmt::DD_NOOP;
  out->dd_method = h.dd_method == fmt::DD_BLOCK ? MGH_DD_BLOCK
                   : h.dd_method == fmt::DD_VARIABLE ? MGH_DD_VARIABLE : MGH_DD_MAXDIM;
  out->dd_dim = h.dd_dim;
  out->dd_size = h.dd_size;
  out->l_target = h.l_target;
  out->reorder = h.reorder ? 1 : 0;
  out->lossless = h.compressor == fmt::COMP_X_HUFFMAN ? MGH_LOSSLESS_HUFFMAN
                  : h.compressor == fmt::COMP_X_HUFFMAN_LZ4 ? MGH_LOSSLESS_HUFFMAN_LZ4
                  : h.compressor == fmt::COMP_X_HUFFMAN_ZSTD ? MGH_LOSSLESS_HUFFMAN_ZSTD : MGH_LOSSLESS_CPU;
  out->huff_dict_size = h.huff_dict_size;
  out->huff_block_size = h.huff_block_size;
  if (metadata_size_out) *metadata_size_out = ms;
  return MGH_SUCCESS;
}

int mgh_huffman_codebook(const uint32_t *freq, uint64_t dict_size, uint64_t *code_out,
                         uint64_t *first_out, uint64_t *entry_out, uint64_t *keys_out) {
  if (!freq || !dict_size || !code_out || !first_out || !entry_out || !keys_out)
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  try {
    const huff::Codebook cb = huff::build_codebook(std::vector<unsigned>(freq, freq + dict_size));
    std::memcpy(code_out, cb.code.data(), dict_size * 8);
    std::memcpy(first_out, cb.first.data(), 64 * 8);
    std::memcpy(entry_out, cb.entry.data(), 64 * 8);
    std::memcpy(keys_out, cb.keys.data(), dict_size * 8);
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_INVALID_ARGUMENT, e.what());
  }
  return MGH_SUCCESS;
}

int mgh_lossless_create(mgh_lossless_ctx **out, int dev_id) {
  if (!out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  *out = new mgh_lossless_ctx();
  (*out)->dev = dev_id;
  return MGH_SUCCESS;
}

void mgh_lossless_destroy(mgh_lossless_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  for (DevBuf *b : {&c->freq, &c->code, &c->bits, &c->entry, &c->total, &c->units, &c->tables, &c->oidx, &c->oval, &c->state, &c->dtable,
                    &c->sync})
    b->release();
  c->pin.release();
  c->dpin.release();
  c->tagpin.release();
  delete c;
}

int mgh_lossless_compress(mgh_lossless_ctx *ctx, const int64_t *d_q, uint64_t n, uint64_t dict,
                          uint64_t chunk, int lossless, int zstd_level, const uint64_t *d_oidx,
                          const int64_t *d_oval, uint64_t ocount, const uint8_t **payload_out,
                          uint64_t *size_out, void *stream) {
  if (!ctx || !d_q || !payload_out || !size_out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (hipSetDevice(ctx->dev) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  try {
    HL_TRY(lossless_compress(ctx, d_q, n, dict, chunk, lossless, zstd_level, d_oidx, d_oval, ocount,
                             (hipStream_t)stream));
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
  if (!ctx->on_host) {
    ctx->host.resize(ctx->lay.total);
    HL_TRY(record_write(ctx, ctx->host.data(), (hipStream_t)stream));
    HL_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  *payload_out = ctx->host.data();
  *size_out = ctx->host.size();
  return MGH_SUCCESS;
}

int mgh_lossless_compress_device(mgh_lossless_ctx *ctx, const int64_t *d_q, uint64_t n, uint64_t dict,
                                 uint64_t chunk, const uint64_t *d_oidx, const int64_t *d_oval,
                                 uint64_t ocount, void *d_record_out, uint64_t capacity, uint64_t *size_out,
                                 void *stream) {
  if (!ctx || !d_q || !d_record_out || !size_out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (hipSetDevice(ctx->dev) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  if (!is_device_pointer(d_record_out)) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "d_record_out must be device memory");
  try {
    HL_TRY(lossless_compress(ctx, d_q, n, dict, chunk, MGH_LOSSLESS_HUFFMAN, 0, d_oidx, d_oval, ocount,
                             (hipStream_t)stream, nullptr, ~(uint64_t)0, 0, false, (uint8_t *)d_record_out,
                             (size_t)capacity));
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
  if (ctx->overflow || ctx->record_size() > capacity)
    return hl_fail(MGH_ERR_OUTPUT_TOO_LARGE, "record does not fit the buffer");
  HL_TRY(record_write(ctx, d_record_out, (hipStream_t)stream));
  HL_HIP(hipStreamSynchronize((hipStream_t)stream));
  *size_out = ctx->record_size();
  return MGH_SUCCESS;
}

int mgh_lossless_decompress(mgh_lossless_ctx *ctx, const uint8_t *payload, uint64_t size, int lossless,
                            int64_t *d_q, uint64_t n, const uint64_t **oidx_out,
                            const int64_t **oval_out, uint64_t *ocount_out, void *stream) {
  if (!ctx || !payload || !d_q || !ocount_out) return hl_fail(MGH_ERR_INVALID_ARGUMENT, "NULL argument");
  if (hipSetDevice(ctx->dev) != hipSuccess) return hl_fail(MGH_ERR_DEVICE, "hipSetDevice");
  try {
    HL_TRY(lossless_decompress(ctx, payload, size, lossless, d_q, n, ocount_out, (hipStream_t)stream));
  } catch (const std::exception &e) {
    return hl_fail(MGH_ERR_DEVICE, e.what());
  }
  if (oidx_out) *oidx_out = (const uint64_t *)ctx->oidx.p;
  if (oval_out) *oval_out = (const int64_t *)ctx->oval.p;
  return MGH_SUCCESS;
}

} // extern "C"
