// Huffman stage of the lossless compressor for gfx950.
//
// Produces / consumes the reference's serialized Huffman payload
// (include/mgard-x/Lossless/ParallelHuffman/Huffman.hpp:163-239, Deserialize :283-370):
//
//   size_t primary_count | int dict_size | int chunk_size | size_t huffmeta_size (= 2 nchunk)
//   size_t bits_per_chunk[nchunk] | size_t word_entry[nchunk]
//   size_t decodebook_size | u8 decodebook[ 8*64 first | 8*64 entry | 8*dict keys ]
//   size_t ddata_size | u64 ddata[ddata_size]           (each member naturally aligned)
//   u64 outlier_count | u64 outlier_idx[count] | i64 outliers[count]
//
// with H = u64 code units filled MSB first, every chunk of `chunk_size` symbols starting on a
// unit boundary (Deflate.hpp:33-76, Condense.hpp), and a canonical code described by
// first[l] / entry[l] / keys[] exactly as Decode.hpp:52-106 consumes it: a code of length l is
// recognised by v >= first[l], its symbol is keys[entry[l] + v - first[l]], unused lengths carry
// first = 2^64-1 (GenerateCW.hpp:77-83). The stock decoder therefore reads these payloads.
//
// How it is built here (not a port of the reference's kernels):
//  * histogram with LDS-privatised bins (k_histogram);
//  * code lengths on the host (build_codebook: two-queue Huffman over the used symbols, codes
//    assigned longest-first so that first[] has the property above; ~0.1 ms);
//  * encoding in ONE pass over the symbols (k_encode_chain): a workgroup per chunk stages symbols
//    and code table in LDS, counts its bits, gets its unit offset by a decoupled look-back over
//    one status word per chunk, assembles the chunk's stream in LDS and writes it out in whole
//    lines at its final (condensed) position. k_chunk_bits + k_unit_offsets + k_encode are the
//    two-pass fallback for parameters whose tables do not fit in LDS;
//  * decoding parallel INSIDE the chunks although the format has entry points per chunk only
//    (k_decode_ring: speculative subsequences that re-synchronise, two-level table built by
//    build_decode_table). k_decode_par (the same without rings / second-level table) and
//    k_decode (one lane per chunk; used below 1024 symbols per chunk) remain as cross-checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <queue>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace mgh {
namespace huff {

constexpr int kUnitBits = 64;
constexpr int kMaxCodeBits = 56;  // codeword representation: length in the top byte

// ---------------------------------------------------------------------------------------
// host: code construction
// ---------------------------------------------------------------------------------------
struct Codebook {
  std::vector<uint64_t> code;  // [dict]: (len << 56) | value, 0 for unused symbols
  std::vector<uint64_t> first, entry;  // [64]
  std::vector<uint64_t> keys;          // [dict]: symbols in decreasing-frequency / code order
  int max_len = 0;
  uint64_t total_bits = 0;  // length of the code stream without the chunks' padding: sum of frequency x code length
  // work arrays of build_codebook (they keep their capacity from call to call: the construction
  // runs once per subdomain on the critical path of mgh_compress)
  std::vector<uint64_t> w_asc, w_weight;
  std::vector<int> w_left, w_right, w_depth, w_len, w_idx;
};

// Symbols by decreasing frequency, ties by DECREASING symbol -- the order the reference arrives at
// (GetCodebook.hpp:44-57,125-128: stable ascending sort by frequency over the symbols in
// increasing order, then the arrays are reversed); code lengths from a two-queue Huffman tree over
// the leaves in increasing weight; canonical codes longest-first (GenerateCW.hpp:54-83).
// 8192 used symbols: ~0.03 ms on one core of the GPU box's host.
inline void build_codebook(const unsigned *freq, int dict, Codebook &cb) {
  cb.max_len = 0;
  cb.total_bits = 0;
  cb.code.assign(dict, 0);
  cb.first.assign(kUnitBits, ~(uint64_t)0);
  cb.entry.assign(kUnitBits, 0);
  cb.keys.resize(dict);
  // used symbols in increasing order as (frequency << 32 | symbol)
  cb.w_asc.resize(2 * (size_t)dict);
  uint64_t *a = cb.w_asc.data(), *t = a + dict;
  constexpr int kDigit = 11, kRadix = 1 << kDigit, kPass = 3;
  int cnt[kRadix + 1];
  std::memset(cnt, 0, sizeof(cnt));
  int nz = 0;
  unsigned differ = 0, f0 = 0;  // bits in which the used counts differ from the first one
  for (int i = 0; i < dict; i++) {
    const unsigned f = freq[i];
    a[nz] = ((uint64_t)f << 32) | (uint32_t)i;
    if (f) {
      if (!nz) f0 = f;
      differ |= f ^ f0;
      cnt[(f & (kRadix - 1)) + 1]++;
      nz++;
    }
  }
  // the unused ones follow the used ones in keys[], in decreasing symbol order (what the reversed
  // stable sort leaves)
  if (nz < dict) {
    int k = nz;
    for (int i = dict - 1; i >= 0; i--)
      if (!freq[i]) cb.keys[k++] = (uint64_t)i;
  }
  if (nz == 0) return;
  {  // stable LSD radix sort by frequency, 11 bits a pass; a digit all counts share is skipped,
     // and a pass counts the digits of the next one while it scatters
    uint64_t *src = a, *dst = t;
    bool counted = true;  // cnt[] holds the digit counts of pass `ps`
    for (int ps = 0; ps < kPass; ps++) {
      const int shift = ps * kDigit;
      if (!((differ >> shift) & (kRadix - 1))) {
        counted = false;
        continue;
      }
      if (!counted) {
        std::memset(cnt, 0, sizeof(cnt));
        for (int i = 0; i < nz; i++) cnt[((src[i] >> (32 + shift)) & (kRadix - 1)) + 1]++;
      }
      for (int k = 0; k < kRadix; k++) cnt[k + 1] += cnt[k];
      const int nshift = shift + kDigit;
      const bool more = ps + 1 < kPass && ((differ >> nshift) & (kRadix - 1));
      if (more) {
        int nxt[kRadix + 1];
        std::memset(nxt, 0, sizeof(nxt));
        for (int i = 0; i < nz; i++) {
          const uint64_t e = src[i];
          dst[cnt[(e >> (32 + shift)) & (kRadix - 1)]++] = e;
          nxt[((e >> (32 + nshift)) & (kRadix - 1)) + 1]++;
        }
        std::memcpy(cnt, nxt, sizeof(cnt));
        counted = true;
      } else {
        for (int i = 0; i < nz; i++) dst[cnt[(src[i] >> (32 + shift)) & (kRadix - 1)]++] = src[i];
        counted = false;
      }
      std::swap(src, dst);
    }
    a = src;  // ascending (frequency, symbol); order[k] of the reference = symbol of a[nz - 1 - k]
  }
  cb.w_len.resize(nz);
  int *len = cb.w_len.data();  // indexed like order[]: k = 0 is the most frequent symbol
  if (nz == 1) {
    len[0] = 1;
  } else {
    // Huffman tree with two queues: leaves 0..nz-1 in increasing weight, inner nodes behind them
    // in the order they are made (their weights are non-decreasing); a leaf wins a tie.
    // (Branches, not selects: the pattern of takes is regular enough to predict, and a select
    // chain through the weights just stored measured five times slower.)
    cb.w_weight.resize(2 * (size_t)nz);
    cb.w_left.resize(2 * (size_t)nz);
    cb.w_right.resize(2 * (size_t)nz);
    cb.w_depth.resize(2 * (size_t)nz);
    uint64_t *w = cb.w_weight.data();
    int *lc = cb.w_left.data(), *rc = cb.w_right.data(), *depth = cb.w_depth.data();
    for (int i = 0; i < nz; i++) w[i] = a[i] >> 32;
    int leaf = 0, inner = nz, end = nz;
    for (int k = 0; k < nz - 1; k++) {
      int x, y;
      if (leaf < nz && (inner >= end || w[leaf] <= w[inner])) x = leaf++; else x = inner++;
      if (leaf < nz && (inner >= end || w[leaf] <= w[inner])) y = leaf++; else y = inner++;
      w[end] = w[x] + w[y];
      lc[end] = x;
      rc[end] = y;
      end++;
    }
    depth[end - 1] = 0;
    for (int i = end - 1; i >= nz; i--) {
      depth[lc[i]] = depth[i] + 1;
      depth[rc[i]] = depth[i] + 1;
    }
    for (int i = 0; i < nz; i++) len[nz - 1 - i] = depth[i];
  }
  // lengths are non-decreasing along order[] up to ties in the tree; canonical assignment needs the
  // symbols grouped by length: counting sort of the order by length (stable)
  // (the lengths come in long runs: the loops below carry a run's counter in a register instead
  // of incrementing through memory element by element)
  int count[kUnitBits + 2] = {0}, at[kUnitBits + 2] = {0};
  for (int i = 0; i < nz;) {
    const int l = len[i];
    int j = i + 1;
    while (j < nz && len[j] == l) j++;
    if (l > kMaxCodeBits) throw std::runtime_error("Huffman: codeword longer than 56 bits");
    count[l] += j - i;
    cb.max_len = l > cb.max_len ? l : cb.max_len;
    i = j;
  }
  cb.w_idx.resize(nz);
  int *idx = cb.w_idx.data();
  for (int l = 1; l <= kUnitBits; l++) at[l + 1] = at[l] + count[l];
  for (int i = 0; i < nz;) {
    const int l = len[i];
    int pos = at[l];
    int j = i;
    while (j < nz && len[j] == l) idx[pos++] = j++;
    at[l] = pos;
    i = j;
  }
  // first[l]: longest codes start at 0, every shorter length continues above the prefixes of
  // the longer ones: first[l] = ceil((first[l+1] + count[l+1]) / 2)
  uint64_t first[kUnitBits + 2] = {0};
  for (int l = cb.max_len - 1; l >= 1; l--) first[l] = (first[l + 1] + (uint64_t)count[l + 1] + 1) / 2;
  // a lone symbol: the reference's canonical code counts up from 0 and is then complemented
  // (GenerateCW.hpp:54-71,209-222), which leaves the one-bit code "1"
  if (nz == 1) first[1] = 1;
  uint64_t e = 0;
  for (int l = 1; l < kUnitBits; l++) {
    cb.entry[l] = e;
    if (l <= cb.max_len && count[l]) cb.first[l] = first[l];
    e += l <= cb.max_len ? (uint64_t)count[l] : 0;
  }
  // keys[] in code order; codes: the j-th symbol of length l (in keys order) gets first[l] + j
  // (keys[] is grouped by length)
  for (int k = 0; k < nz;) {
    const int l = len[idx[k]];
    uint64_t v = ((uint64_t)l << kMaxCodeBits) | first[l];
    const int k1 = k + count[l];
    for (; k < k1; k++) {
      const uint32_t sym = (uint32_t)a[nz - 1 - idx[k]];
      cb.keys[k] = (uint64_t)sym;
      cb.code[sym] = v++;
      cb.total_bits += (uint64_t)l * freq[sym];
    }
  }
}

inline Codebook build_codebook(const std::vector<unsigned> &freq) {
  Codebook cb;
  build_codebook(freq.data(), (int)freq.size(), cb);
  return cb;
}

// ---------------------------------------------------------------------------------------
// device kernels
// ---------------------------------------------------------------------------------------
// Histogram of the symbols (int64 values in [0, dict)); bins privatised in LDS. Launch with
// kHistThreads threads and few workgroups (two per CU): every workgroup ends with one global atomic
// per used bin, and with 2048 workgroups of 256 those 16 M atomics on 8192 addresses were a third
// of the kernel's time.
constexpr int kHistThreads = 512;
template <typename SYM>  // int64_t (the reference's quantized array) or uint16_t symbols
__global__ void __launch_bounds__(kHistThreads)
k_histogram(const SYM *__restrict__ q, size_t n, int dict, unsigned *__restrict__ freq) {
  extern __shared__ unsigned bins[];
  for (int i = threadIdx.x; i < dict; i += kHistThreads) bins[i] = 0;
  __syncthreads();
  const size_t nth = (size_t)gridDim.x * kHistThreads;
  size_t done = 0;
  if (sizeof(SYM) == 2 && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {  // 8 symbols per load
    const size_t nv = n / 8;
    const uint4 *qv = reinterpret_cast<const uint4 *>(q);
    auto count8 = [&](const uint4 &v) {
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned a = w[k] & 0xffffu, b = w[k] >> 16;
        if (a < (unsigned)dict) atomicAdd(&bins[a], 1u);
        if (b < (unsigned)dict) atomicAdd(&bins[b], 1u);
      }
    };
    // four loads in flight per lane
    size_t i = (size_t)blockIdx.x * kHistThreads + threadIdx.x;
    for (; i + 3 * nth < nv; i += 4 * nth) {
      const uint4 v0 = qv[i], v1 = qv[i + nth], v2 = qv[i + 2 * nth], v3 = qv[i + 3 * nth];
      count8(v0);
      count8(v1);
      count8(v2);
      count8(v3);
    }
    for (; i < nv; i += nth) count8(qv[i]);
    done = nv * 8;
  }
  for (size_t i = done + (size_t)blockIdx.x * kHistThreads + threadIdx.x; i < n; i += nth) {
    const uint64_t s = (uint64_t)q[i];
    if (s < (uint64_t)dict) atomicAdd(&bins[s], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < dict; i += kHistThreads)
    if (bins[i]) atomicAdd(&freq[i], bins[i]);
}

// bits of every chunk (one workgroup per chunk)
__global__ void __launch_bounds__(256)
k_chunk_bits(const int64_t *__restrict__ q, size_t n, int chunk, const uint64_t *__restrict__ code,
             unsigned long long *__restrict__ bits) {
  const size_t base = (size_t)blockIdx.x * chunk;
  const size_t cnt = min((size_t)chunk, n - base);
  unsigned long long s = 0;
  for (size_t i = threadIdx.x; i < cnt; i += 256) s += code[q[base + i]] >> kMaxCodeBits;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ unsigned long long w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bits[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// entry[c] = sum_{k<c} ceil(bits[k] / 64); total units in *total (one workgroup)
__global__ void __launch_bounds__(1024)
k_unit_offsets(const unsigned long long *__restrict__ bits, size_t nchunk,
               unsigned long long *__restrict__ entry, unsigned long long *__restrict__ total) {
  __shared__ unsigned long long part[1024];
  const size_t per = (nchunk + 1023) / 1024;
  const size_t lo = min(nchunk, threadIdx.x * per), hi = min(nchunk, lo + per);
  unsigned long long s = 0;
  for (size_t i = lo; i < hi; i++) s += (bits[i] + 63) / 64;
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned long long v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned long long run = part[threadIdx.x] - s;
  for (size_t i = lo; i < hi; i++) {
    entry[i] = run;
    run += (bits[i] + 63) / 64;
  }
  if (threadIdx.x == 1023) *total = part[1023];
}

// Write the bit stream of every chunk at its final position (units pre-zeroed). One workgroup
// per chunk; thread t packs the symbols [t*run, (t+1)*run) of the chunk.
__global__ void __launch_bounds__(256)
k_encode(const int64_t *__restrict__ q, size_t n, int chunk, const uint64_t *__restrict__ code,
         const unsigned long long *__restrict__ entry, unsigned long long *__restrict__ out) {
  const size_t base = (size_t)blockIdx.x * chunk;
  const size_t cnt = min((size_t)chunk, n - base);
  const size_t run = (cnt + 255) / 256;
  const size_t lo = min(cnt, threadIdx.x * run), hi = min(cnt, lo + run);
  unsigned long long s = 0;
  for (size_t i = lo; i < hi; i++) s += code[q[base + i]] >> kMaxCodeBits;
  // exclusive scan of the per-thread bit counts
  __shared__ unsigned long long sc[256];
  sc[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const unsigned long long v = threadIdx.x >= (unsigned)off ? sc[threadIdx.x - off] : 0;
    __syncthreads();
    sc[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned long long pos = sc[threadIdx.x] - s;  // first bit of this thread's run
  if (lo >= hi || s == 0) return;
  unsigned long long *dst = out + entry[blockIdx.x];
  size_t w = pos / kUnitBits;
  int room = kUnitBits - (int)(pos % kUnitBits);  // free bits in the current unit
  unsigned long long acc = 0;
  bool first_unit = true;
  auto flush = [&](bool last) {
    // the first and the last unit of a run may be shared with the neighbouring runs
    if (first_unit || last) atomicOr(&dst[w], acc);
    else dst[w] = acc;
    first_unit = false;
  };
  for (size_t i = lo; i < hi; i++) {
    const uint64_t c = code[q[base + i]];
    const int len = (int)(c >> kMaxCodeBits);
    const unsigned long long val = c & (((uint64_t)1 << kMaxCodeBits) - 1);
    if (len <= room) {
      room -= len;
      acc |= val << room;
      if (room == 0) {
        flush(false);
        w++;
        acc = 0;
        room = kUnitBits;
      }
    } else {
      const int rest = len - room;
      acc |= val >> rest;
      flush(false);
      w++;
      room = kUnitBits - rest;
      acc = val << room;
    }
  }
  if (room != kUnitBits) flush(true);
}

constexpr int kEncRun = 40;      // symbols per thread the single-pass encoder keeps in registers (default chunk: 20480 = 40 x 512)
constexpr int kEncThreads = 512;  // threads of an encoder workgroup (two workgroups per CU with 32-bit code entries)


// One pass over the symbols instead of two (k_chunk_bits + k_encode): the unit offset of a
// chunk is the sum of the unit counts of all chunks before it, which a workgroup obtains while
// its symbols sit in LDS by a decoupled look-back over a status word per chunk (the single-pass
// prefix scan of Merrill & Garland): chunks are handed out in ticket order, so every
// predecessor of a running workgroup is running or done and the wait is bounded; a workgroup
// publishes (AGGREGATE | its own units) as soon as it has counted its bits, and
// (INCLUSIVE | units up to and including itself) once it knows its offset. Status words, ticket
// are read and written with device-scope atomic loads / stores (the XCDs do not share an L2;
// read-modify-writes instead serialise in the L2 channel of a status line that up to 64
// successors poll: 0.73 -> see DESIGN.md).
//   state: [0] ticket counter, [1] total units (written by the last chunk), [2] overflow flag,
//          [3 + c] status of chunk c; all zero before the launch.
// A workgroup zeroes its units before packing (threads meet in shared units with atomicOr);
// nothing is written beyond cap_units: the overflow flag tells the caller that the stream did not
// fit (bits[] / entry[] / total are still right).
constexpr unsigned long long kStAggregate = 1ull << 62, kStInclusive = 2ull << 62,
                             kStValue = (1ull << 62) - 1;

// CODE: uint64_t entries (length << 56 | value) or, when no code is longer than 27 bits, uint32_t
// entries (length << 27 | value): half the LDS, two workgroups per CU instead of one.
// A code longer than 27 bits belongs to one of the rarest symbols; a histogram of 10^9 symbols has
// a few dozen of them. They do not force the 64-bit table on the kernel: their 32-bit entry is an
// ESCAPE (length field 31, value = index) into a list of 64-bit entries behind the table.
template <typename CODE> struct CodeEntry;
template <> struct CodeEntry<uint64_t> { static constexpr int shift = kMaxCodeBits; };
template <> struct CodeEntry<uint32_t> { static constexpr int shift = 27; };
constexpr int kShortCodeBits = 27;
constexpr unsigned kEscapeLen = 31;  // length field of an escape entry

// (length, value) of a table entry; escapes resolved through `slong` (32-bit tables only)
template <typename CODE>
__device__ __forceinline__ void entry_decode(CODE c, const unsigned long long *slong, int &len,
                                             unsigned long long &val) {
  constexpr int SH = CodeEntry<CODE>::shift;
  len = (int)(c >> SH);
  val = (unsigned long long)(c & (((CODE)1 << SH) - 1));
  if (sizeof(CODE) == 4 && len == (int)kEscapeLen) {
    const unsigned long long e = slong[val];
    len = (int)(e >> kMaxCodeBits);
    val = e & (((unsigned long long)1 << kMaxCodeBits) - 1);
  }
}
template <typename CODE>
__device__ __forceinline__ unsigned entry_len(CODE c, const unsigned long long *slong) {
  constexpr int SH = CodeEntry<CODE>::shift;
  unsigned len = (unsigned)(c >> SH);
  if (sizeof(CODE) == 4 && len == kEscapeLen)
    len = (unsigned)(slong[c & (((CODE)1 << SH) - 1)] >> kMaxCodeBits);
  return len;
}

// dynamic LDS of k_encode_chain (bytes)
inline size_t encode_chain_lds(size_t dict, size_t entry_bytes, size_t chunk) {
  return (dict * entry_bytes + 15) / 16 * 16 + chunk * 2;
}

// Code units at an address that need not be 8-byte aligned: a record inside a container starts
// wherever the header and the records before it end, and the encoder / decoder work on the units
// where they lie (gfx950 serves unaligned global dwordx2 accesses; atomics need alignment).
constexpr int kSyncLanes = 64;  // synchronisation points per chunk = lanes of the decoder's wave
__device__ __forceinline__ unsigned load_u32(const unsigned *p, size_t i) {  // (any byte alignment)
  unsigned v;
  __builtin_memcpy(&v, reinterpret_cast<const unsigned char *>(p) + 4 * i, 4);
  return v;
}
__device__ __forceinline__ unsigned long long load_unit(const unsigned long long *p, size_t i) {
  unsigned long long v;
  __builtin_memcpy(&v, reinterpret_cast<const unsigned char *>(p) + 8 * i, 8);
  return v;
}
__device__ __forceinline__ void store_unit(unsigned long long *p, size_t i, unsigned long long v) {
  __builtin_memcpy(reinterpret_cast<unsigned char *>(p) + 8 * i, &v, 8);
}

// (two workgroups per CU with 32-bit code entries = four waves per SIMD: 128 VGPRs)
template <typename SYM, typename CODE>
__global__ void __launch_bounds__(kEncThreads, sizeof(CODE) == 4 ? 4 : 2)
k_encode_chain(const SYM *__restrict__ q, size_t n, int chunk, int dict, size_t nchunk,
               const CODE *__restrict__ code, unsigned long long *__restrict__ state,
               unsigned long long *__restrict__ bits, unsigned long long *__restrict__ entry,
               unsigned long long *__restrict__ out, unsigned long long cap_units, int nlong,
               unsigned *__restrict__ sync, unsigned long long sync_tag) {
  constexpr int SH = CodeEntry<CODE>::shift;
  // LDS: code table [dict] | symbols of the chunk (encode_chain_lds() is the host's copy of this
  // layout). The escape list stays in global memory behind the table (8-byte aligned): its entries
  // belong to symbols that occur a handful of times in the whole subdomain.
  extern __shared__ unsigned long long enc_lds[];
  CODE *scode = reinterpret_cast<CODE *>(enc_lds);
  unsigned short *ssym = reinterpret_cast<unsigned short *>(
      reinterpret_cast<unsigned char *>(enc_lds) + ((size_t)dict * sizeof(CODE) + 15) / 16 * 16);
  const unsigned long long *slong = reinterpret_cast<const unsigned long long *>(
      reinterpret_cast<const unsigned char *>(code) + ((size_t)dict * sizeof(CODE) + 7) / 8 * 8);
  (void)nlong;
  __shared__ unsigned sc[kEncThreads];   // bit offset of every thread's run in the chunk
  __shared__ unsigned swave[kEncThreads / 64];
  __shared__ unsigned long long sh_id, sh_entry;
  if (threadIdx.x == 0) sh_id = atomicAdd(&state[0], 1ull);
  for (int i = threadIdx.x; i < dict; i += kEncThreads) scode[i] = code[i];
  __syncthreads();
  const size_t id = (size_t)sh_id;
  const size_t base = id * chunk;
  const size_t cnt = min((size_t)chunk, n - base);
  size_t staged = 0;
  if (sizeof(SYM) == 2 && (reinterpret_cast<uintptr_t>(q + base) & 15) == 0) {
    // 16-bit symbols: straight 16-byte copies
    const uint4 *src = reinterpret_cast<const uint4 *>(q + base);
    uint4 *dst4 = reinterpret_cast<uint4 *>(ssym);
    for (size_t i = threadIdx.x; i < cnt / 8; i += kEncThreads) dst4[i] = src[i];
    staged = cnt / 8 * 8;
  }
  for (size_t i = staged + threadIdx.x; i < cnt; i += kEncThreads) ssym[i] = (unsigned short)q[base + i];
  __syncthreads();
  const size_t run = (cnt + kEncThreads - 1) / kEncThreads;
  const size_t lo = min(cnt, threadIdx.x * run), hi = min(cnt, lo + run);
  // The code entries of the run are kept in registers between the counting and the packing
  // (default chunk: 40 symbols per thread): the table look-ups of a thread are then independent
  // LDS reads in flight together, and the packing loop runs from registers. Longer runs (bigger
  // chunks) read the table twice.
  constexpr int RR = kEncRun;
  CODE cc[RR];
  const bool in_regs = run <= (size_t)RR;
  unsigned s = 0;
  // Fast path: a full chunk of the default size with a 32-bit table (every thread has exactly RR
  // symbols). The generic loop below tests `lo + k < hi` per symbol, which puts every look-up into
  // a basic block of its own -- symbol load, wait, table load, wait, 40 times in a row. Here the
  // thread's 80 bytes of symbols arrive as five 16-byte LDS reads (lane stride 80 bytes = 20 banks:
  // conflict-free) and the 40 table look-ups are issued back to back. Escape entries (codes longer
  // than 27 bits: the rarest symbols) only raise a flag; a thread that holds one recounts and packs
  // on the generic path.
  const bool fast = sizeof(CODE) == 4 && cnt == (size_t)RR * kEncThreads;
  bool esc = false;
  if (fast) {
    const uint4 *sp = reinterpret_cast<const uint4 *>(ssym + (size_t)threadIdx.x * RR);
    unsigned wv[RR / 2];
#pragma unroll
    for (int j = 0; j < RR / 8; j++) {
      const uint4 t4 = sp[j];
      wv[4 * j + 0] = t4.x;
      wv[4 * j + 1] = t4.y;
      wv[4 * j + 2] = t4.z;
      wv[4 * j + 3] = t4.w;
    }
#pragma unroll
    for (int k = 0; k < RR; k++) cc[k] = scode[(k & 1) ? (wv[k / 2] >> 16) : (wv[k / 2] & 0xffffu)];
#pragma unroll
    for (int k = 0; k < RR; k++) {
      const unsigned l = (unsigned)(cc[k] >> SH);
      esc |= l == kEscapeLen;
      s += l;
    }
    if (esc) {  // (unrolled: a loop would index cc[] dynamically and send the array to scratch)
      s = 0;
#pragma unroll
      for (int k = 0; k < RR; k++) s += entry_len(cc[k], slong);
    }
  } else if (in_regs) {
#pragma unroll
    for (int k = 0; k < RR; k++) {
      cc[k] = lo + k < hi ? scode[ssym[lo + k]] : (CODE)0;  // (a zero entry packs nothing)
      s += entry_len(cc[k], slong);
    }
  } else {
    for (size_t i = lo; i < hi; i++) s += entry_len(scode[ssym[i]], slong);
  }
  // exclusive scan of the per-thread bit counts: inside the waves with shuffles, across them
  // through swave[] (a chunk holds fewer than 2^32 bits: the host limits the chunk size)
  unsigned incl = s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) swave[wv] = incl;
  __syncthreads();
  unsigned wave_off = 0, total_bits = 0;
  for (int k = 0; k < kEncThreads / 64; k++) {
    const unsigned t = swave[k];
    if (k < wv) wave_off += t;
    total_bits += t;
  }
  sc[threadIdx.x] = wave_off + incl;  // inclusive, as below
  const unsigned long long chunk_bits = total_bits;
  const unsigned long long my_units = (chunk_bits + kUnitBits - 1) / kUnitBits;
  if (threadIdx.x < 64) {  // wave 0: publish, look back, publish
    const int lane = threadIdx.x;
    unsigned long long excl = 0;
    if (id > 0) {
      if (lane == 0) __hip_atomic_store(&state[3 + id], kStAggregate | my_units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long p = (long long)id - 1;  // first predecessor this round looks at
      while (true) {
        const long long mine = p - lane;
        unsigned long long st = kStInclusive;  // lanes before chunk 0 terminate the walk
        if (mine >= 0) st = __hip_atomic_load(&state[3 + mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ready = __ballot((st >> 62) != 0);
        const unsigned long long incl = __ballot((st >> 62) == 2);
        // usable prefix of lanes: all ready up to (and including) the first inclusive one
        const int first_unready = ready == ~0ull ? 64 : __ffsll((long long)~ready) - 1;
        const int first_incl = incl ? __ffsll((long long)incl) - 1 : 64;
        const int upto = min(first_unready, first_incl + 1);  // lanes [0, upto) are added
        unsigned long long v = lane < upto && mine >= 0 ? (st & kStValue) : 0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        excl += __shfl(v, 0, 64);
        if (first_incl < upto) break;  // reached an inclusive prefix
        p -= upto;                     // (upto == 0: spin on the same predecessor)
        if (upto == 0) __builtin_amdgcn_s_sleep(1);
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&state[3 + id], kStInclusive | (excl + my_units), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bits[id] = chunk_bits;
      entry[id] = excl;
      if (id == nchunk - 1) state[1] = excl + my_units;
      if (excl + my_units > cap_units) atomicOr(&state[2], 1ull);
      sh_entry = excl;
    }
  }
  __syncthreads();
  const unsigned long long e0 = sh_entry;
  if (e0 + my_units > cap_units) return;
  unsigned long long *dst = out + e0;
  const unsigned pos = sc[threadIdx.x] - s;  // first bit of this thread's run
  // Synchronisation points for the decoder (kSyncLanes per chunk; not part of the reference's
  // payload: the caller stores them behind it). The decoder cuts a chunk's stream into 64
  // subsequences of B = ceil(bits / 64) bits, one per lane; entry k says where the first code
  // that STARTS at or behind bit k * B begins (delta = its distance from k * B) and which symbol
  // of the chunk it is. With them a lane decodes its symbols once and straight to their place;
  // without them it has to find its first code boundary by decoding speculatively, and the
  // symbols' places by a counting pass (k_decode_ring). A boundary k * B in (start, end] of a
  // code makes the NEXT symbol the entry of k; entries of subsequences in which no code starts
  // keep (0, cnt): no symbols.
  if (sync) {  // (same decision in every thread; in_regs by the host's choice of chunk size)
    __shared__ unsigned ssync[kSyncLanes];
    if (threadIdx.x < kSyncLanes) ssync[threadIdx.x] = threadIdx.x ? (unsigned)cnt : 0u;
    __syncthreads();
    const unsigned B = (total_bits + kSyncLanes - 1) / kSyncLanes;
    if (B && in_regs && s) {
      // boundaries in (pos, pos + s]: none for seven threads of eight, one for the others (a run is
      // much shorter than a subsequence)
      const unsigned k_lo = pos / B + 1, k_hi = min((pos + s) / B, (unsigned)kSyncLanes - 1);
      if (fast && !esc && k_lo == k_hi) {
        // ... and the symbol behind it by straight-line selects over the lengths in registers
        const unsigned nb = k_lo * B;
        unsigned o = pos, idx = 0, dl = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < RR; k++) {
          o += (unsigned)(cc[k] >> SH);  // end of symbol lo + k = start of symbol lo + k + 1
          const bool hit = !found && o >= nb;
          idx = hit ? (unsigned)(k + 1) : idx;
          dl = hit ? o - nb : dl;
          found |= hit;
        }
        if (found && lo + idx < cnt) ssync[k_lo] = (min(dl, 0xffffu) << 16) | (unsigned)(lo + idx);
      } else if (k_lo <= k_hi) {  // (several boundaries in one run, a partial chunk, escape entries, 64-bit table: from the LDS copies)
        unsigned o = pos, kk = k_lo, nb = k_lo * B;
        for (size_t i = lo; i < hi && kk <= k_hi; i++) {
          o += entry_len(scode[ssym[i]], slong);
          while (o >= nb && kk <= k_hi) {
            if (i + 1 < cnt) ssync[kk] = (min(o - nb, 0xffffu) << 16) | (unsigned)(i + 1);
            kk++;
            nb += B;
          }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < kSyncLanes) sync[id * kSyncLanes + threadIdx.x] = ssync[threadIdx.x];
    // (the section's tag in front of the entries: the caller copies tag + entries in one piece)
    if (id == 0 && threadIdx.x == 0) {
      sync[-2] = (unsigned)(sync_tag & 0xffffffffu);
      sync[-1] = (unsigned)(sync_tag >> 32);
    }
  }
  const bool active = lo < hi && s != 0;
  // pack the runs into `buf` (zeroed): whole units are stored, the first and the last unit of a
  // run are shared with the neighbours and OR-ed in
  auto pack = [&](unsigned long long *buf) {
    size_t w = pos / kUnitBits;
    int room = kUnitBits - (int)(pos % kUnitBits);
    unsigned long long acc = 0;
    bool first_unit = true;
    auto flush = [&](bool last) {
      if (first_unit || last) atomicOr(&buf[w], acc);
      else buf[w] = acc;
      first_unit = false;
    };
    auto put = [&](CODE c) {
      int len;
      unsigned long long val;
      entry_decode(c, slong, len, val);
      if (len <= room) {
        room -= len;
        acc |= val << room;
        if (room == 0) {
          flush(false);
          w++;
          acc = 0;
          room = kUnitBits;
        }
      } else {
        const int rest = len - room;
        acc |= val >> rest;
        flush(false);
        w++;
        room = kUnitBits - rest;
        acc = val << room;
      }
    };
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < RR; k++) put(cc[k]);
    } else {
      for (size_t i = lo; i < hi; i++) put(scode[ssym[i]]);
    }
    if (room != kUnitBits) flush(true);
  };
  // With the code entries in registers the symbol area of the LDS is free: the chunk's stream is
  // assembled there (LDS atomics, no zeroing of global memory) and written out in whole lines --
  // packing straight into global memory stores 8 bytes per lane from a few lanes at a time
  // (0.48 of 0.65 ms at 512^3). Streams longer than that area (more than 16 bits per symbol on
  // average) take the direct path.
  const bool via_lds = in_regs && my_units * 8 <= (unsigned long long)chunk * 2;
  if (via_lds) {
    unsigned long long *obuf = reinterpret_cast<unsigned long long *>(ssym);
    for (size_t i = threadIdx.x; i < my_units; i += kEncThreads) obuf[i] = 0;
    __syncthreads();
    if (fast && !esc) {
      // Branch-light packing in 32-bit words: codes of at most 27 bits are appended to a 64-bit
      // accumulator that never holds 32 or more bits between two symbols (31 + 27 < 64), a full
      // upper half is OR-ed into the stream (word w of the MSB-first stream is the 32-bit half
      // w ^ 1 of the little-endian 64-bit units). Every word goes out by ds_or, so the words a
      // run shares with its neighbours need no special case.
      unsigned *o32 = reinterpret_cast<unsigned *>(obuf);
      unsigned wi = pos >> 5, fill = pos & 31;
      unsigned long long acc = 0;
#pragma unroll
      for (int k = 0; k < RR; k++) {
        const unsigned len = (unsigned)(cc[k] >> SH);
        const unsigned long long val = (unsigned long long)(cc[k] & (((CODE)1 << SH) - 1));
        acc |= val << (64 - fill - len);
        fill += len;
        if (fill >= 32) {
          atomicOr(&o32[wi ^ 1], (unsigned)(acc >> 32));
          acc <<= 32;
          fill -= 32;
          wi++;
        }
      }
      if (fill) atomicOr(&o32[wi ^ 1], (unsigned)(acc >> 32));
    } else if (active) {
      pack(obuf);
    }
    __syncthreads();
    for (size_t i = threadIdx.x; i < my_units; i += kEncThreads) store_unit(dst, i, obuf[i]);
  } else {
    // (atomics on the destination: an unaligned one cannot take this path -- state[2] bit 1 asks
    // the host to run the encoder again into an aligned buffer; rare: more than 16 bits per symbol)
    if (reinterpret_cast<uintptr_t>(out) & 7) {
      if (threadIdx.x == 0) atomicOr(&state[2], 2ull);
      return;
    }
    for (size_t i = threadIdx.x; i < my_units; i += kEncThreads) dst[i] = 0;
    __syncthreads();
    if (active) pack(dst);
  }
}

// Canonical decoding, one lane per chunk (the chunks are the only entry points of the stream);
// semantics of Decode.hpp:52-106: a code of length l is recognised by v >= first[l] and stands
// for keys[entry[l] + v - first[l]]. Codes of up to `tb` bits are resolved with one
// lookup in a prefix table the workgroup builds in LDS from first / entry / keys (lengths with
// first = 2^64-1 are unused, the number of codes of a length is entry[l+1] - entry[l]); longer
// codes continue bit by bit from the same 64-bit window. `units` must be readable one element
// past the stream (the window peeks ahead).
// The table has 2^tb entries (tb chosen by the host so that table + keys fit in LDS: 15 bits
// for the default dictionary); dynamic LDS = 4 * 2^tb + 2 * dict (rounded up to 8) + 8 KiB.
__global__ void __launch_bounds__(64)
k_decode(const unsigned long long *__restrict__ units, const unsigned long long *__restrict__ bits,
         const unsigned long long *__restrict__ entry_of_chunk, size_t nchunk, int chunk, size_t n,
         int dict, int tb, const unsigned long long *__restrict__ first,
         const unsigned long long *__restrict__ entry, const unsigned long long *__restrict__ keys,
         int64_t *__restrict__ q) {
  __shared__ unsigned long long sfirst[64], sentry[64];
  extern __shared__ unsigned dyn_lds[];
  unsigned *table = dyn_lds;  // (length << 16) | symbol, 0 = longer code
  unsigned short *skeys = reinterpret_cast<unsigned short *>(dyn_lds + (1u << tb));  // [dict]
  // [16][64] staged code units, behind the keys (8-byte aligned)
  unsigned long long *ring = reinterpret_cast<unsigned long long *>(
      dyn_lds + (1u << tb) + (((unsigned)dict * 2 + 7) / 8) * 2);
  sfirst[threadIdx.x] = first[threadIdx.x];
  sentry[threadIdx.x] = entry[threadIdx.x];
  for (unsigned i = threadIdx.x; i < (1u << tb); i += 64) table[i] = 0;
  for (int i = threadIdx.x; i < dict; i += 64) skeys[i] = (unsigned short)keys[i];
  __syncthreads();
  // (longest codes first: should a foreign stream's entry[] make a length look longer than it
  // is, the surplus slots are rewritten by the shorter, valid codes that own them)
  for (int l = tb; l >= 1; l--) {
    if (sfirst[l] != ~0ull && sentry[l] < (unsigned long long)dict) {
      unsigned long long cnt = (l + 1 < 64 ? sentry[l + 1] : (unsigned long long)dict) - sentry[l];
      cnt = min(cnt, (unsigned long long)dict - sentry[l]);  // (damaged tables must not spin here)
      cnt = min(cnt, 1ull << l);
      const unsigned span = 1u << (tb - l);
      // codes first[l] .. first[l] + cnt - 1, each covering `span` table slots
      for (unsigned long long j = threadIdx.x; j < cnt * span; j += 64) {
        const unsigned long long code = sfirst[l] + j / span;
        const unsigned long long slot = (code << (tb - l)) + j % span;
        const unsigned long long k = sentry[l] + j / span;
        if (slot < (1ull << tb) && k < (unsigned long long)dict)
          table[slot] = ((unsigned)l << 16) | (unsigned)skeys[k];
      }
    }
    __syncthreads();
  }
  const size_t c0 = (size_t)blockIdx.x * 64 + threadIdx.x;
  const bool have = c0 < nchunk;  // (idle lanes stay in the loop: it uses wave votes)
  const size_t c = have ? c0 : nchunk - 1;
  const unsigned long long *src = units + entry_of_chunk[c];
  const unsigned long long total = have ? bits[c] : 0;
  int64_t *dst = q + c * (size_t)chunk;
  const size_t cap = min((size_t)chunk, n - c * (size_t)chunk);
  const unsigned long long nun = (total + 63) / 64;  // (src[nun] exists: the stream is padded)
  // Code units reach the lanes through a per-lane ring in LDS that ALL lanes top up together,
  // kRing loads in flight each: a wave waits for its youngest outstanding load, and nearly every
  // iteration some lane crosses a unit boundary, so a load per crossing would stall the wave
  // every iteration (measured: 0.7 us per symbol step).
  constexpr int kRing = 16;
  const int lane = threadIdx.x;
  unsigned long long filled = 0;  // units [0, filled) of this lane's stream have been staged
  auto refill = [&](unsigned long long cw) {
    unsigned long long tmp[kRing];
#pragma unroll
    for (int k = 0; k < kRing; k++) {
      const unsigned long long idx = filled + k;
      tmp[k] = (idx < cw + kRing && idx <= nun) ? src[idx] : 0;
    }
#pragma unroll
    for (int k = 0; k < kRing; k++) {
      const unsigned long long idx = filled + k;
      if (idx < cw + kRing && idx <= nun) ring[(idx % kRing) * 64 + lane] = tmp[k];
    }
    filled = min(cw + kRing, nun + 1);
  };
  unsigned long long i = 0, cw = 0;
  refill(0);
  unsigned long long cur = ring[lane], nxt = ring[64 + lane];
  size_t produced = 0;
  bool live = total > 0;
  while (__any(live)) {
    // invariant at the top: units cw .. cw+2 are staged (one crossing can happen below)
    if (__any(live && cw + 2 >= filled && filled <= nun)) refill(cw);
    if (live) {
      const int sh = (int)(i & 63);
      const unsigned long long win = sh ? (cur << sh) | (nxt >> (64 - sh)) : cur;
      const unsigned e = table[win >> (64 - tb)];
      int l = 0;
      unsigned long long sym = 0;
      bool hit = e != 0;
      if (hit) {
        l = (int)(e >> 16);
        sym = e & 0xffff;
      } else {
        for (l = tb + 1; l <= 56; l++) {
          const unsigned long long v = win >> (64 - l);
          if (v >= sfirst[l]) {
            const unsigned long long k = sentry[l] + v - sfirst[l];
            if (k < (unsigned long long)dict) {
              sym = skeys[k];
              hit = true;
            }
            break;
          }
        }
      }
      // (a miss or an overrun means a corrupt stream: stop instead of indexing out of range)
      if (!hit || i + l > total) {
        live = false;
      } else {
        dst[produced++] = (int64_t)sym;
        i += l;
        if ((i >> 6) != cw) {
          cw++;
          cur = nxt;
          nxt = ring[((cw + 1) % kRing) * 64 + lane];
        }
        live = i < total && produced < cap;
      }
    }
  }
}

// Parallel decoding INSIDE a chunk (the serial kernel above needs 20480 dependent steps per
// chunk). One wave per chunk, 16 chunks per workgroup sharing the LDS tables. The chunk's bit
// stream is cut into 64 equal subsequences, one per lane:
//   1. every lane decodes its subsequence speculatively from the cut (lane 0's start is a true
//      code boundary, the others probably are not) and remembers where it ended;
//   2. a lane whose start differs from the end of its left neighbour restarts from that end;
//      repeated until no lane changes. Huffman streams re-synchronise within a few symbols, so
//      in practice one or two rounds; at most 64 (every round fixes at least the first wrong
//      lane), and then every start is a true boundary;
//   3. symbol counts are prefix-summed across the wave and every lane decodes its subsequence
//      once more, now writing the symbols to their final positions.
// Same tables and semantics as k_decode. Dynamic LDS as for k_decode without the unit ring.
constexpr int kParWaves = 16;
constexpr int kParBatch = 32;  // symbols a lane decodes between two write-outs

__global__ void __launch_bounds__(64 * kParWaves)
k_decode_par(const unsigned long long *__restrict__ units, const unsigned long long *__restrict__ bits,
             const unsigned long long *__restrict__ entry_of_chunk, size_t nchunk, int chunk, size_t n,
             int dict, int tb, const unsigned long long *__restrict__ first,
             const unsigned long long *__restrict__ entry, const unsigned long long *__restrict__ keys,
             int64_t *__restrict__ q) {
  __shared__ unsigned long long sfirst[64], sentry[64];
  extern __shared__ unsigned dyn_lds[];
  unsigned *table = dyn_lds;
  unsigned short *skeys = reinterpret_cast<unsigned short *>(dyn_lds + (1u << tb));
  // per wave: kParBatch symbols of every lane, staged for the coalesced write-out
  unsigned short *stage = skeys + (((size_t)dict + 3) / 4 * 4) + (size_t)(threadIdx.x >> 6) * (64 * kParBatch);
  constexpr int NT = 64 * kParWaves;
  if (threadIdx.x < 64) {
    sfirst[threadIdx.x] = first[threadIdx.x];
    sentry[threadIdx.x] = entry[threadIdx.x];
  }
  for (unsigned i = threadIdx.x; i < (1u << tb); i += NT) table[i] = 0;
  for (int i = threadIdx.x; i < dict; i += NT) skeys[i] = (unsigned short)keys[i];
  __syncthreads();
  // (longest codes first: should a foreign stream's entry[] make a length look longer than it
  // is, the surplus slots are rewritten by the shorter, valid codes that own them)
  for (int l = tb; l >= 1; l--) {
    if (sfirst[l] != ~0ull && sentry[l] < (unsigned long long)dict) {
      unsigned long long cnt = (l + 1 < 64 ? sentry[l + 1] : (unsigned long long)dict) - sentry[l];
      cnt = min(cnt, (unsigned long long)dict - sentry[l]);
      cnt = min(cnt, 1ull << l);
      const unsigned span = 1u << (tb - l);
      for (unsigned long long j = threadIdx.x; j < cnt * span; j += NT) {
        const unsigned long long code = sfirst[l] + j / span;
        const unsigned long long slot = (code << (tb - l)) + j % span;
        const unsigned long long k = sentry[l] + j / span;
        if (slot < (1ull << tb) && k < (unsigned long long)dict)
          table[slot] = ((unsigned)l << 16) | (unsigned)skeys[k];
      }
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const size_t c = (size_t)blockIdx.x * kParWaves + (threadIdx.x >> 6);
  if (c >= nchunk) return;  // (whole wave; no block-wide barrier follows)
  const unsigned long long *src = units + entry_of_chunk[c];
  const unsigned long long total = bits[c];
  int64_t *dst = q ? q + c * (size_t)chunk : nullptr;
  const unsigned long long cap = min((size_t)chunk, n - c * (size_t)chunk);
  const unsigned long long nun = (total + 63) / 64;
  const unsigned long long B = (total + 63) / 64;  // bits per subsequence (ceil(total / 64))
  const unsigned long long lim = min(((unsigned long long)lane + 1) * B, total);

  // decode from bit `start` while the position is below `lim`; optionally store the symbols at
  // out[0 .. max_out). Returns the end position, *count = symbols decoded.
  auto run = [&](unsigned long long start, unsigned long long *count, int64_t *out,
                 unsigned long long max_out) {
    unsigned long long pos = start, cnt = 0;
    if (pos >= lim) {
      *count = 0;
      return pos;
    }
    unsigned long long cw = pos >> 6;
    unsigned long long cur = src[min(cw, nun)], nxt = src[min(cw + 1, nun)];
    while (pos < lim) {
      const int sh = (int)(pos & 63);
      const unsigned long long win = sh ? (cur << sh) | (nxt >> (64 - sh)) : cur;
      const unsigned e = table[win >> (64 - tb)];
      int l = 0;
      unsigned long long sym = 0;
      bool hit = e != 0;
      if (hit) {
        l = (int)(e >> 16);
        sym = e & 0xffff;
      } else {
        for (l = tb + 1; l <= 56; l++) {
          const unsigned long long v = win >> (64 - l);
          if (v >= sfirst[l]) {
            const unsigned long long k = sentry[l] + v - sfirst[l];
            if (k < (unsigned long long)dict) {
              sym = skeys[k];
              hit = true;
            }
            break;
          }
        }
      }
      if (!hit || pos + l > total) {  // corrupt stream (or a speculative start that runs off the
        pos = total;                  // end): give up on this subsequence
        break;
      }
      if (out && cnt < max_out) out[cnt] = (int64_t)sym;
      cnt++;
      pos += l;
      if ((pos >> 6) != cw) {
        cw++;
        cur = nxt;
        nxt = src[min(cw + 1, nun)];
      }
    }
    *count = cnt;
    return pos;
  };

  unsigned long long s = min((unsigned long long)lane * B, total), cnt = 0;
  unsigned long long e = run(s, &cnt, nullptr, 0);
  for (int it = 0; it < 64; it++) {
    unsigned long long pe = __shfl_up(e, 1, 64);
    if (lane == 0) pe = 0;
    const bool changed = s != pe;
    if (!__any(changed)) break;
    if (changed) {
      s = pe;
      e = run(s, &cnt, nullptr, 0);
    }
  }
  // exclusive prefix sum of the symbol counts
  unsigned long long off = cnt;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long v = __shfl_up(off, d, 64);
    if (lane >= d) off += v;
  }
  off -= cnt;
  // 3. final pass: every lane decodes its subsequence again, kParBatch symbols at a time into
  // LDS; after each batch the wave writes the 64 runs out one after the other, 8-byte elements
  // of a run side by side (a lane storing its own symbols directly touches 64 different cache
  // lines per store instruction: 1.1 of 2.9 ms at 512^3).
  if (!dst) return;
  unsigned long long pos = s, done = 0;  // bit position, symbols of this lane written so far
  unsigned long long cw = pos >> 6;
  unsigned long long cur = src[min(cw, nun)], nxt = src[min(cw + 1, nun)];
  bool live = pos < lim && cnt > 0;
  while (__any(live)) {
    int got = 0;
    while (live && got < kParBatch) {
      const int sh = (int)(pos & 63);
      const unsigned long long win = sh ? (cur << sh) | (nxt >> (64 - sh)) : cur;
      const unsigned e = table[win >> (64 - tb)];
      int l = 0;
      unsigned sym = 0;
      bool hit = e != 0;
      if (hit) {
        l = (int)(e >> 16);
        sym = e & 0xffff;
      } else {
        for (l = tb + 1; l <= 56; l++) {
          const unsigned long long v = win >> (64 - l);
          if (v >= sfirst[l]) {
            const unsigned long long k = sentry[l] + v - sfirst[l];
            if (k < (unsigned long long)dict) {
              sym = skeys[k];
              hit = true;
            }
            break;
          }
        }
      }
      if (!hit || pos + l > total) {
        live = false;
        break;
      }
      stage[lane * kParBatch + got] = (unsigned short)sym;
      got++;
      pos += l;
      if ((pos >> 6) != cw) {
        cw++;
        cur = nxt;
        nxt = src[min(cw + 1, nun)];
      }
      live = pos < lim && done + got < cnt;
    }
    // write-out (LDS traffic of one wave: in order, no barrier needed)
    for (int L = 0; L < 64; L++) {
      const int n_L = __shfl(got, L, 64);
      const unsigned long long o_L = __shfl(off + done, L, 64);
      if (lane < n_L && o_L + lane < cap) dst[o_L + lane] = (int64_t)stage[L * kParBatch + lane];
    }
    done += got;
  }
}

// ---------------------------------------------------------------------------------------
// Two-level decoding table (host), built from the decodebook of the payload.
//   root: 2^tb entries indexed by the next tb bits of the stream;
//     (len << 16) | symbol          a code of len <= tb bits (replicated over its 2^(tb-len) slots)
//     1<<31 | sub_bits << 24 | off  codes longer than tb bits with this prefix: second lookup at
//                                   table[off + next sub_bits bits], entries (len << 16) | symbol
//     0                             neither (unused prefix, or left to the comparison path of the
//                                   kernel: sub-table wider than kSubBitsMax or table budget spent)
// Codes are taken longest first, as the kernels do, so that should a foreign stream's entry[]
// make a length look longer than it is, the shorter valid codes overwrite the surplus.
// ---------------------------------------------------------------------------------------
constexpr int kSubBitsMax = 11;

inline std::vector<uint32_t> build_decode_table(const uint64_t *first, const uint64_t *entry,
                                                const uint64_t *keys, int dict, int tb,
                                                size_t max_entries) {
  std::vector<uint32_t> t((size_t)1 << tb, 0);
  auto count_of = [&](int l) -> uint64_t {
    if (first[l] == ~(uint64_t)0 || entry[l] >= (uint64_t)dict) return 0;
    uint64_t cnt = (l + 1 < 64 ? entry[l + 1] : (uint64_t)dict) - entry[l];
    if (l + 1 < 64 && entry[l + 1] < entry[l]) cnt = 0;
    cnt = std::min<uint64_t>(cnt, (uint64_t)dict - entry[l]);
    cnt = std::min<uint64_t>(cnt, (uint64_t)1 << l);
    if (first[l] >= ((uint64_t)1 << l)) return 0;
    return std::min<uint64_t>(cnt, ((uint64_t)1 << l) - first[l]);
  };
  // pass 1: widest code under every root prefix that has long codes
  std::vector<uint8_t> group_len((size_t)1 << tb, 0);
  for (int l = tb + 1; l <= kMaxCodeBits; l++) {
    const uint64_t cnt = count_of(l);
    for (uint64_t j = 0; j < cnt; j++) {
      const uint64_t prefix = (first[l] + j) >> (l - tb);
      group_len[prefix] = (uint8_t)std::max<int>(group_len[prefix], l);
    }
  }
  for (size_t pfx = 0; pfx < group_len.size(); pfx++) {
    if (!group_len[pfx]) continue;
    const int sb = group_len[pfx] - tb;
    if (sb > kSubBitsMax || t.size() + ((size_t)1 << sb) > max_entries) continue;  // comparison path
    t[pfx] = 0x80000000u | ((uint32_t)sb << 24) | (uint32_t)t.size();
    t.resize(t.size() + ((size_t)1 << sb), 0);
  }
  // pass 2: fill, longest codes first
  for (int l = kMaxCodeBits; l >= 1; l--) {
    const uint64_t cnt = count_of(l);
    for (uint64_t j = 0; j < cnt; j++) {
      const uint64_t code = first[l] + j;
      const uint32_t val = ((uint32_t)l << 16) | (uint32_t)(keys[entry[l] + j] & 0xffff);
      if (l <= tb) {
        const uint64_t lo = code << (tb - l);
        for (uint64_t k = 0; k < ((uint64_t)1 << (tb - l)); k++) t[lo + k] = val;
      } else {
        const uint32_t r = t[code >> (l - tb)];
        if (!(r & 0x80000000u)) continue;
        const int sb = (int)((r >> 24) & 0x7f);
        const size_t off = r & 0xffffff;
        const uint64_t rest = code & (((uint64_t)1 << (l - tb)) - 1);  // the l - tb bits after the prefix
        const uint64_t lo = rest << (sb - (l - tb));
        for (uint64_t k = 0; k < ((uint64_t)1 << (sb - (l - tb))); k++) t[off + lo + k] = val;
      }
    }
  }
  return t;
}

// Parallel decoding inside a chunk like k_decode_par, with two changes that matter:
//  * the code units reach a lane through a per-lane LDS ring that the whole wave tops up at
//    wave-uniform points (every kRingStep symbols; the loads land in registers while the next
//    symbols are decoded and are committed to the ring afterwards). In k_decode_par a lane loads
//    its next unit when it crosses into a new one: lanes cross at different symbols, so nearly
//    every iteration of the wave contains a load and the wave-wide wait on it;
//  * every symbol is resolved with one or two table lookups (build_decode_table): with 64 lanes a
//    wave almost always holds SOME lane with a code longer than the prefix table, so a search
//    over the lengths costs the whole wave on nearly every symbol. The comparison path (slim[],
//    below) remains for prefixes the table leaves out.
// A lane that would run past the units it holds (codes longer than 32 bits on average) pauses
// until the next refill.
//   dynamic LDS: 4 * table_entries | per wave: kRingUnits * 64 * 8 (ring) + 64 * kRingBatch * 2
//   (write-out staging); blockDim.x = 64 * waves.
constexpr int kRingUnits = 8;   // units per lane in the ring
constexpr int kRingFetch = 4;   // units per lane per refill
constexpr int kRingStep = 8;    // symbols between two refills
constexpr int kRingBatch = 16;  // symbols a lane stages between two write-outs (multiple of kRingStep)
// Row length of a lane's staging slots: ONE MORE than the batch. With rows of 16 two-byte slots (32
// bytes) the 64 lanes of the per-symbol store `stage[lane * 16 + got]` fall on four LDS banks, a
// 16-way conflict on every symbol step. Rows of 17 slots spread the lanes over all banks, and the
// write-out's reads stay contiguous. (Measured: no difference at 512^3 -- the decoder is bound by
// neither this nor its instruction count alone, see k_decode_lean and profiles/NOTES.md round 6.)
constexpr int kStageStride = kRingBatch + 1;
constexpr int kRecStride = 8;   // every kRecStride-th code boundary of the first pass is remembered

inline size_t decode_ring_lds(size_t table_entries, int waves) {
  return (table_entries * 4 + 7) / 8 * 8 + (size_t)waves * (kRingUnits * 64 * 8 + 64 * kStageStride * 2);
}

template <typename OUT>  // int64_t (the reference's array of quantized values) or uint16_t symbols
__global__ void __launch_bounds__(1024)
k_decode_ring(const unsigned long long *__restrict__ units, const unsigned long long *__restrict__ bits,
              const unsigned long long *__restrict__ entry_of_chunk, size_t nchunk, int chunk, size_t n,
              int dict, int tb, const unsigned *__restrict__ g_table, unsigned table_entries,
              const unsigned long long *__restrict__ first,
              const unsigned long long *__restrict__ entry, const unsigned long long *__restrict__ keys,
              OUT *__restrict__ q, const unsigned *__restrict__ sync) {
  __shared__ unsigned long long sfirst[64], sentry[64], slim[64];
  __shared__ int smaxlen;
  extern __shared__ unsigned dyn_lds[];
  unsigned *table = dyn_lds;
  const int nwaves = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long *rings = reinterpret_cast<unsigned long long *>(dyn_lds + (table_entries + 1) / 2 * 2);
  unsigned long long *ring = rings + (size_t)wave * (kRingUnits * 64);
  unsigned short *stage = reinterpret_cast<unsigned short *>(rings + (size_t)nwaves * (kRingUnits * 64)) +
                          (size_t)wave * (64 * kStageStride);
  if (threadIdx.x < 64) {
    sfirst[threadIdx.x] = first[threadIdx.x];
    sentry[threadIdx.x] = entry[threadIdx.x];
  }
  for (unsigned i = threadIdx.x; i < table_entries; i += blockDim.x) table[i] = g_table[i];
  __syncthreads();
  // Comparison path for what the table leaves out: with lim[l] = first[l] << (64 - l)
  // (left-aligned lower bound of the codes of length l, 2^64-1 for unused lengths) a window w
  // holds a code of the smallest l with w >= lim[l]; slim[l] = min over (tb, l] of lim makes
  // that a count, len = tb + 1 + #{l in (tb, maxlen] : w < slim[l]} -- no dependent search.
  if (threadIdx.x == 0) {
    unsigned long long m = ~0ull;
    int mx = 0;
    for (int l = 1; l < 64; l++) {
      const bool used = sfirst[l] != ~0ull && l <= kMaxCodeBits;
      if (used) mx = l;
      if (l > tb) {
        if (used) m = min(m, sfirst[l] << (64 - l));
        slim[l] = m;
      } else {
        slim[l] = ~0ull;
      }
    }
    slim[0] = ~0ull;
    smaxlen = mx;
  }
  __syncthreads();
  const int maxlen = smaxlen;
  const size_t c = (size_t)blockIdx.x * nwaves + wave;
  if (c >= nchunk) return;  // (whole wave; no block-wide barrier follows)
  const unsigned long long *src = units + entry_of_chunk[c];
  // (bit positions inside a chunk fit 32 bits: the host sends chunks of at most 2^24 symbols here)
  const unsigned total = (unsigned)min(bits[c], (unsigned long long)chunk * kMaxCodeBits);
  OUT *dst = q + c * (size_t)chunk;
  const unsigned cap = (unsigned)min((size_t)chunk, n - c * (size_t)chunk);
  const unsigned nun = (total + 63) / 64;  // src[nun] is readable (the window peeks ahead)
  const unsigned B = (total + 63) / 64;    // bits per subsequence
  const unsigned lim = min((unsigned)(lane + 1) * B, total);
  unsigned short *rec = stage + lane * kStageStride;  // this lane's slots of the staging area

  // One pass over this lane's subsequence from bit `start`; all lanes of the wave call it
  // together. MODE 0: count the symbols up to position lim, remember where every kRecStride-th of the
  // first kRingBatch * kRecStride codes ended (rec[], relative to start). MODE 1: the same from a corrected start, but stop as
  // soon as a position is reached that the remembered pass went through -- from there on the two
  // are the same pass (Huffman streams re-synchronise within a few symbols), so its end and its
  // remaining count are inherited. MODE 2: decode `want` symbols and write them to dst[out0...].
  // Returns the end position, *count = symbols.
  auto pass = [&](auto mode_tag, unsigned start, unsigned *count, unsigned want, unsigned out0,
                  unsigned rec_start, unsigned rec_end, unsigned rec_cnt) {
    constexpr int MODE = decltype(mode_tag)::value;
    unsigned pos = start, cnt = 0;
    bool live = MODE == 2 ? (pos < lim && want > 0) : pos < lim;
    if (MODE == 1 && live && pos == rec_start) {  // (starts on the remembered pass)
      *count = rec_cnt;
      pos = rec_end;
      live = false;
    }
    // Bit buffer: `win` holds the next `avail` bits of the stream left-aligned (the rest zero);
    // it is topped up 32 bits at a time from the ring (word w = half of unit w / 2, the high half
    // first), so that more than 32 bits are there whenever a code is looked up -- the host sends
    // streams with longer codes to k_decode_par. `wnext` = next word to take.
    const unsigned cw0 = pos >> 6;
    unsigned hi = cw0;
#pragma unroll
    for (int r = 0; r < kRingUnits; r++)  // initial fill of the ring
      ring[((cw0 + r) % kRingUnits) * 64 + lane] = load_unit(src, min(cw0 + r, nun));
    hi = cw0 + kRingUnits;
    unsigned long long win;
    unsigned avail, wnext;
    {
      const unsigned long long u0 = ring[(cw0 % kRingUnits) * 64 + lane];
      const unsigned long long u1 = ring[((cw0 + 1) % kRingUnits) * 64 + lane];
      const int sh = (int)(pos & 63);
      win = sh ? (u0 << sh) | (u1 >> (64 - sh)) : u0;  // 64 valid bits from pos on
      avail = 64;
      wnext = ((pos + 64) >> 5);                        // first word not (completely) in win ...
      // ... pos + 64 is in the middle of that word unless pos is a multiple of 32: keep only the
      // whole words, drop the partial one so that refills stay word-aligned
      const unsigned part = (pos + 64) & 31;
      avail -= part;
      win = part ? (win >> part) << part : win;
    }
    unsigned long long pf[kRingFetch];
    bool pending = false;
    int jrec = 0;           // MODE 1: next remembered boundary to compare with
    bool inherited = !live && MODE == 1 && pos == rec_end && *count == rec_cnt && start == rec_start;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < kRingBatch; k++) rec[k] = 0xffff;
    }
    while (__any(live)) {
      int got = 0;
      for (int rep = 0; rep < kRingBatch / kRingStep; rep++) {
        if (live && !pending) {  // request the next units; they are committed kRingStep symbols later
#pragma unroll
          for (int j = 0; j < kRingFetch; j++) pf[j] = load_unit(src, min(hi + j, nun));
          pending = true;
        }
        for (int k = 0; k < kRingStep; k++) {
          if (live && avail <= 32 && (wnext >> 1) < hi) {  // top up (a lane out of units pauses)
            const unsigned long long u = ring[((wnext >> 1) % kRingUnits) * 64 + lane];
            const unsigned wd = (wnext & 1) ? (unsigned)u : (unsigned)(u >> 32);
            win |= (unsigned long long)wd << (32 - avail);
            avail += 32;
            wnext++;
          }
          if (live && avail > 32) {
            unsigned e = table[win >> (64 - tb)];
            if (e & 0x80000000u) {  // second level: the next sub_bits bits
              const int sb = (int)((e >> 24) & 0x7f);
              e = table[(e & 0xffffff) + (unsigned)((win << tb) >> (64 - sb))];
            }
            int l = (int)(e >> 16);
            unsigned sym = e & 0xffff;
            bool hit = e != 0;
            if (!hit) {  // not in the table
              l = tb + 1;
              for (int j = tb + 1; j <= maxlen; j++) l += win < slim[j] ? 1 : 0;
              if (l <= maxlen) {
                const unsigned long long v = win >> (64 - l);
                const unsigned long long kk = sentry[l] + (v - sfirst[l]);
                if (v >= sfirst[l] && kk < (unsigned long long)dict) {
                  sym = (unsigned)keys[kk] & 0xffff;
                  hit = true;
                }
              }
            }
            if (!hit || pos + l > total) {  // corrupt stream, or a speculative start running off the end
              pos = total;
              live = false;
            } else {
              if (MODE == 2) stage[lane * kStageStride + got] = (unsigned short)sym;
              got++;
              pos += l;
              win <<= l;
              avail -= l;
              cnt++;
              if (MODE == 0 && cnt % kRecStride == 0 && cnt / kRecStride <= (unsigned)kRingBatch)
                rec[cnt / kRecStride - 1] = (unsigned short)min(pos - start, 0xfffeu);
              live = MODE == 2 ? (pos < lim && cnt < want) : pos < lim;
              if (MODE == 1 && live && pos >= rec_start) {
                const unsigned rel = pos - rec_start;
                if (rel == 0) {
                  cnt += rec_cnt;
                  pos = rec_end;
                  live = false;
                  inherited = true;
                } else if (rel < 0xfffeu) {
                  while (jrec < kRingBatch && rec[jrec] < rel) jrec++;
                  if (jrec < kRingBatch && rec[jrec] == rel) {
                    cnt += rec_cnt - (unsigned)(jrec + 1) * kRecStride;
                    pos = rec_end;
                    live = false;
                    inherited = true;
                  }
                }
              }
            }
          }
        }
        // refill point: commit the units requested before these symbols if the ring has room
        if (pending && hi + kRingFetch <= (wnext >> 1) + kRingUnits) {
#pragma unroll
          for (int j = 0; j < kRingFetch; j++) ring[((hi + j) % kRingUnits) * 64 + lane] = pf[j];
          hi += kRingFetch;
          pending = false;
        }
      }
      if (MODE == 2) {  // write-out: 64 / kRingBatch runs per instruction, the elements of a run side by side
        constexpr int RPI = 64 / kRingBatch;  // runs per store instruction
        const unsigned my_o = out0 + cnt - got;
#pragma unroll 4
        for (int it = 0; it < 64 / RPI; it++) {
          const int L = it * RPI + lane / kRingBatch, k = lane % kRingBatch;
          const int n_L = __shfl(got, L, 64);
          const unsigned o_L = __shfl(my_o, L, 64);
          if (k < n_L && o_L + k < cap) dst[o_L + k] = (OUT)stage[L * kStageStride + k];
        }
      }
    }
    if (!(MODE == 1 && inherited && start == rec_start)) *count = cnt;
    return pos;
  };
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>;

  if (sync) {
    // the encoder's synchronisation points (k_encode_chain): start and place of every lane's
    // symbols are given, one decoding pass. (Damaged entries decode garbage inside the chunk's
    // own range of units and symbols at worst: every access below is bounded by total and cap.)
    const unsigned ent = load_u32(sync, c * kSyncLanes + lane);
    const unsigned first_sym = lane ? min(ent & 0xffffu, cap) : 0u;
    unsigned next_sym = __shfl_down(first_sym, 1, 64);
    if (lane == 63) next_sym = cap;
    const unsigned s0 = lane ? (unsigned)min((unsigned long long)lane * B + (ent >> 16), (unsigned long long)total) : 0u;
    const unsigned want = next_sym > first_sym ? next_sym - first_sym : 0u;
    unsigned c2 = 0;
    (void)pass(M2{}, s0, &c2, want, first_sym, 0, 0, 0);
    return;
  }
  unsigned s = min((unsigned)lane * B, total), cnt = 0;
  unsigned e = pass(M0{}, s, &cnt, 0, 0, 0, 0, 0);
  const unsigned rs = s, re = e, rc = cnt;  // the remembered pass
  for (int it = 0; it < 64; it++) {
    unsigned pe = __shfl_up(e, 1, 64);
    if (lane == 0) pe = 0;
    const bool changed = s != pe;
    if (!__any(changed)) break;
    // (all lanes take part in the pass; the unchanged ones with an empty range)
    unsigned c1 = 0;
    const unsigned e1 = pass(M1{}, changed ? pe : total, &c1, 0, 0, rs, re, rc);
    if (changed) {
      s = pe;
      e = e1;
      cnt = c1;
    }
  }
  unsigned off = cnt;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned v = __shfl_up(off, d, 64);
    if (lane >= d) off += v;
  }
  off -= cnt;
  unsigned c2 = 0;
  (void)pass(M2{}, s, &c2, off < cap ? cnt : 0, off, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// Decoder for records WITH synchronisation points (round 6). k_decode_ring at 512^3 is bound by
// VALU issue: 137 instructions per symbol step, one symbol per step. With the encoder's entries
// every lane knows its first code and the place of its symbols, so nothing of the speculative
// machinery is needed, and -- codes of a quantized field are short (4-5 bits on average at 1e-3) --
// most root-table slots hold TWO complete codes: the root entries are pairs,
//     word 0: the entry of build_decode_table (code <= tb bits, pointer to a second level, or 0)
//     word 1: (len1 + len2) << 16 | symbol2 when a second code of len2 <= tb - len1 bits follows
//             the first inside the tb-bit window, else 0,
// and a step that finds word 1 set (and has two symbols left to decode) emits both. The ring of
// code units, the refill points and the write-out are k_decode_ring's.
//   table layout (32-bit words): [2^tb pairs][second-level tables of build_decode_table]
// ---------------------------------------------------------------------------------------
inline std::vector<uint32_t> make_pair_table(const std::vector<uint32_t> &t, int tb) {
  const size_t R = (size_t)1 << tb;
  std::vector<uint32_t> out(2 * R + (t.size() - R), 0);
  for (size_t i = 0; i < R; i++) {
    const uint32_t e1 = t[i];
    uint32_t w0 = e1, w1 = 0;
    if (e1 & 0x80000000u) {
      w0 = (e1 & 0xff000000u) | ((e1 & 0xffffffu) + (uint32_t)R);  // (the second levels moved up by R words)
    } else if (e1) {
      const uint32_t l1 = e1 >> 16;
      if (l1 < (uint32_t)tb) {
        const uint32_t e2 = t[(i << l1) & (R - 1)];  // the window behind the first code, unknown bits zero
        if (e2 && !(e2 & 0x80000000u)) {
          const uint32_t l2 = e2 >> 16;
          if (l1 + l2 <= (uint32_t)tb) w1 = ((l1 + l2) << 16) | (e2 & 0xffffu);  // (all of its bits are real)
        }
      }
    }
    out[2 * i] = w0;
    out[2 * i + 1] = w1;
  }
  for (size_t k = R; k < t.size(); k++) out[2 * R + (k - R)] = t[k];
  return out;
}

constexpr int kSyncSteps = 8;  // decoding steps (one or two symbols each) between two refills / write-outs

template <typename OUT>
__global__ void __launch_bounds__(1024)
k_decode_sync(const unsigned long long *__restrict__ units, const unsigned long long *__restrict__ bits,
              const unsigned long long *__restrict__ entry_of_chunk, size_t nchunk, int chunk, size_t n,
              int dict, int tb, const unsigned *__restrict__ g_table, unsigned table_words,
              const unsigned long long *__restrict__ first,
              const unsigned long long *__restrict__ entry, const unsigned long long *__restrict__ keys,
              OUT *__restrict__ q, const unsigned *__restrict__ sync) {
  __shared__ unsigned long long sfirst[64], sentry[64], slim[64];
  __shared__ int smaxlen;
  extern __shared__ unsigned dyn_lds[];
  unsigned *table = dyn_lds;
  const int nwaves = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long *rings = reinterpret_cast<unsigned long long *>(dyn_lds + (table_words + 1) / 2 * 2);
  unsigned long long *ring = rings + (size_t)wave * (kRingUnits * 64);
  unsigned short *stage = reinterpret_cast<unsigned short *>(rings + (size_t)nwaves * (kRingUnits * 64)) +
                          (size_t)wave * (64 * kStageStride);
  if (threadIdx.x < 64) {
    sfirst[threadIdx.x] = first[threadIdx.x];
    sentry[threadIdx.x] = entry[threadIdx.x];
  }
  for (unsigned i = threadIdx.x; i < table_words; i += blockDim.x) table[i] = g_table[i];
  __syncthreads();
  if (threadIdx.x == 0) {  // comparison path for prefixes the table leaves out (see k_decode_ring)
    unsigned long long m = ~0ull;
    int mx = 0;
    for (int l = 1; l < 64; l++) {
      const bool used = sfirst[l] != ~0ull && l <= kMaxCodeBits;
      if (used) mx = l;
      if (l > tb) {
        if (used) m = min(m, sfirst[l] << (64 - l));
        slim[l] = m;
      } else {
        slim[l] = ~0ull;
      }
    }
    slim[0] = ~0ull;
    smaxlen = mx;
  }
  __syncthreads();
  const int maxlen = smaxlen;
  const size_t c = (size_t)blockIdx.x * nwaves + wave;
  if (c >= nchunk) return;  // (whole wave; no block-wide barrier follows)
  const unsigned long long *src = units + entry_of_chunk[c];
  const unsigned total = (unsigned)min(bits[c], (unsigned long long)chunk * kMaxCodeBits);
  OUT *dst = q + c * (size_t)chunk;
  const unsigned cap = (unsigned)min((size_t)chunk, n - c * (size_t)chunk);
  const unsigned nun = (total + 63) / 64;  // src[nun] is readable (the window peeks ahead)
  const unsigned B = (total + 63) / 64;    // bits per subsequence
  const unsigned lim = min((unsigned)(lane + 1) * B, total);
  // start and place of this lane's symbols (damaged entries decode garbage inside the chunk's own
  // range of units and symbols at worst: every access below is bounded by total and cap)
  const unsigned ent = load_u32(sync, c * kSyncLanes + lane);
  const unsigned first_sym = lane ? min(ent & 0xffffu, cap) : 0u;
  unsigned next_sym = __shfl_down(first_sym, 1, 64);
  if (lane == 63) next_sym = cap;
  const unsigned s0 = lane ? (unsigned)min((unsigned long long)lane * B + (ent >> 16), (unsigned long long)total) : 0u;
  const unsigned want = next_sym > first_sym ? next_sym - first_sym : 0u;

  unsigned pos = s0, cnt = 0;
  bool live = pos < lim && want > 0;
  const unsigned cw0 = pos >> 6;
#pragma unroll
  for (int r = 0; r < kRingUnits; r++)  // initial fill of the ring
    ring[((cw0 + r) % kRingUnits) * 64 + lane] = load_unit(src, min(cw0 + r, nun));
  unsigned hi = cw0 + kRingUnits;
  unsigned long long win;
  unsigned avail, wnext;
  {
    const unsigned long long u0 = ring[(cw0 % kRingUnits) * 64 + lane];
    const unsigned long long u1 = ring[((cw0 + 1) % kRingUnits) * 64 + lane];
    const int sh = (int)(pos & 63);
    win = sh ? (u0 << sh) | (u1 >> (64 - sh)) : u0;  // 64 valid bits from pos on
    avail = 64;
    wnext = ((pos + 64) >> 5);
    const unsigned part = (pos + 64) & 31;  // (keep whole words only: refills stay word-aligned)
    avail -= part;
    win = part ? (win >> part) << part : win;
  }
  unsigned long long pf[kRingFetch];
  bool pending = false;
  while (__any(live)) {
    int got = 0;
    if (live && !pending) {  // request the next units; they are committed kSyncSteps steps later
#pragma unroll
      for (int j = 0; j < kRingFetch; j++) pf[j] = load_unit(src, min(hi + j, nun));
      pending = true;
    }
#pragma unroll
    for (int k = 0; k < kSyncSteps; k++) {
      if (live && avail <= 32 && (wnext >> 1) < hi) {  // top up (a lane out of units pauses)
        const unsigned long long u = ring[((wnext >> 1) % kRingUnits) * 64 + lane];
        const unsigned wd = (wnext & 1) ? (unsigned)u : (unsigned)(u >> 32);
        win |= (unsigned long long)wd << (32 - avail);
        avail += 32;
        wnext++;
      }
      if (live && avail > 32) {
        const uint2 pr = *reinterpret_cast<const uint2 *>(table + 2 * (unsigned)(win >> (64 - tb)));
        const unsigned l2 = pr.y >> 16;
        if (pr.y != 0 && cnt + 2 <= want && pos + l2 <= total) {  // two codes inside the window
          stage[lane * kStageStride + got] = (unsigned short)(pr.x & 0xffffu);
          stage[lane * kStageStride + got + 1] = (unsigned short)(pr.y & 0xffffu);
          got += 2;
          cnt += 2;
          pos += l2;
          win <<= l2;
          avail -= l2;
          live = pos < lim && cnt < want;
        } else {
          unsigned e = pr.x;
          if (e & 0x80000000u) {  // second level: the next sub_bits bits
            const int sb = (int)((e >> 24) & 0x7f);
            e = table[(e & 0xffffff) + (unsigned)((win << tb) >> (64 - sb))];
          }
          int l = (int)(e >> 16);
          unsigned sym = e & 0xffff;
          bool hit = e != 0;
          if (!hit) {  // not in the table
            l = tb + 1;
            for (int j = tb + 1; j <= maxlen; j++) l += win < slim[j] ? 1 : 0;
            if (l <= maxlen) {
              const unsigned long long v = win >> (64 - l);
              const unsigned long long kk = sentry[l] + (v - sfirst[l]);
              if (v >= sfirst[l] && kk < (unsigned long long)dict) {
                sym = (unsigned)keys[kk] & 0xffff;
                hit = true;
              }
            }
          }
          if (!hit || pos + l > total) {  // corrupt stream
            pos = total;
            live = false;
          } else {
            stage[lane * kStageStride + got] = (unsigned short)sym;
            got++;
            cnt++;
            pos += l;
            win <<= l;
            avail -= l;
            live = pos < lim && cnt < want;
          }
        }
      }
    }
    // refill point: commit the units requested before these steps if the ring has room
    if (pending && hi + kRingFetch <= (wnext >> 1) + kRingUnits) {
#pragma unroll
      for (int j = 0; j < kRingFetch; j++) ring[((hi + j) % kRingUnits) * 64 + lane] = pf[j];
      hi += kRingFetch;
      pending = false;
    }
    {  // write-out: 64 / kRingBatch runs per instruction, the elements of a run side by side
      constexpr int RPI = 64 / kRingBatch;  // runs per store instruction
      const unsigned my_o = first_sym + cnt - got;
#pragma unroll 4
      for (int it = 0; it < 64 / RPI; it++) {
        const int L = it * RPI + lane / kRingBatch, k = lane % kRingBatch;
        const int n_L = __shfl(got, L, 64);
        const unsigned o_L = __shfl(my_o, L, 64);
        if (k < n_L && o_L + k < cap) dst[o_L + k] = (OUT)stage[L * kStageStride + k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_decode_ring's writing pass for records with synchronisation points, with the per-symbol step
// rewritten WITHOUT divergent control flow (round 6). The ring decoder is bound by instruction issue:
// 137 ISA instructions per symbol step at 512^3, about half of them the scalar bookkeeping of a dozen
// `if`s per step (s_and_saveexec / s_cbranch_execz / s_or exec around a handful of vector
// instructions each). Here a step is straight-line code under selects -- top-up word always read,
// table always looked up, the staged symbol always written (into a slot that only counts when the
// step was good) -- and everything rare (second table level, prefixes the table leaves out, a
// damaged stream) sits behind WAVE-UNIFORM branches (`__any`), which cost one scalar branch when no
// lane needs them. Same table, ring, refill points and write-out as k_decode_ring; same symbols.
// RESULT: 75 instead of ~135 static instructions per step and the SAME kernel time (512^3 record,
// 16-bit symbols out: 516 vs 511 us). Ablations (MGH_HUFF_DBG, tools/exp_decode_kernels.sh): without
// the global stores 458 us, with a constant 9-bit code instead of the table's answer 416 us, without
// the write-out loop 429 us, without both 309 us -- a 300 us skeleton (ring traffic, top-ups, the
// step's bookkeeping on four waves per SIMD) that neither variant touches. Kept as the instrumented
// kernel (MGH_HUFF_LEAN=1); k_decode_ring stays the default.
// ---------------------------------------------------------------------------------------
template <typename OUT>
__global__ void __launch_bounds__(1024)
k_decode_lean(const unsigned long long *__restrict__ units, const unsigned long long *__restrict__ bits,
              const unsigned long long *__restrict__ entry_of_chunk, size_t nchunk, int chunk, size_t n,
              int dict, int tb, const unsigned *__restrict__ g_table, unsigned table_entries,
              const unsigned long long *__restrict__ first,
              const unsigned long long *__restrict__ entry, const unsigned long long *__restrict__ keys,
              OUT *__restrict__ q, const unsigned *__restrict__ sync) {
  const int dbg = tb >> 8;  // (developer: 1 = no stores, 2 = constant code instead of the table's, 4 = no write-out)
  tb &= 0xff;
  __shared__ unsigned long long sfirst[64], sentry[64], slim[64];
  __shared__ int smaxlen;
  extern __shared__ unsigned dyn_lds[];
  unsigned *table = dyn_lds;
  const int nwaves = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long *rings = reinterpret_cast<unsigned long long *>(dyn_lds + (table_entries + 1) / 2 * 2);
  unsigned long long *ring = rings + (size_t)wave * (kRingUnits * 64);
  const unsigned *ring32 = reinterpret_cast<const unsigned *>(ring);
  unsigned short *stage = reinterpret_cast<unsigned short *>(rings + (size_t)nwaves * (kRingUnits * 64)) +
                          (size_t)wave * (64 * kStageStride);
  if (threadIdx.x < 64) {
    sfirst[threadIdx.x] = first[threadIdx.x];
    sentry[threadIdx.x] = entry[threadIdx.x];
  }
  for (unsigned i = threadIdx.x; i < table_entries; i += blockDim.x) table[i] = g_table[i];
  __syncthreads();
  if (threadIdx.x == 0) {  // comparison path for prefixes the table leaves out (see k_decode_ring)
    unsigned long long m = ~0ull;
    int mx = 0;
    for (int l = 1; l < 64; l++) {
      const bool used = sfirst[l] != ~0ull && l <= kMaxCodeBits;
      if (used) mx = l;
      if (l > tb) {
        if (used) m = min(m, sfirst[l] << (64 - l));
        slim[l] = m;
      } else {
        slim[l] = ~0ull;
      }
    }
    slim[0] = ~0ull;
    smaxlen = mx;
  }
  __syncthreads();
  const int maxlen = smaxlen;
  const size_t c = (size_t)blockIdx.x * nwaves + wave;
  if (c >= nchunk) return;  // (whole wave; no block-wide barrier follows)
  const unsigned long long *src = units + entry_of_chunk[c];
  const unsigned total = (unsigned)min(bits[c], (unsigned long long)chunk * kMaxCodeBits);
  OUT *dst = q + c * (size_t)chunk;
  const unsigned cap = (unsigned)min((size_t)chunk, n - c * (size_t)chunk);
  const unsigned nun = (total + 63) / 64;  // src[nun] is readable (the window peeks ahead)
  const unsigned B = (total + 63) / 64;    // bits per subsequence
  const unsigned lim = min((unsigned)(lane + 1) * B, total);
  const unsigned ent = load_u32(sync, c * kSyncLanes + lane);
  const unsigned first_sym = lane ? min(ent & 0xffffu, cap) : 0u;
  unsigned next_sym = __shfl_down(first_sym, 1, 64);
  if (lane == 63) next_sym = cap;
  const unsigned s0 = lane ? (unsigned)min((unsigned long long)lane * B + (ent >> 16), (unsigned long long)total) : 0u;
  const unsigned want = next_sym > first_sym ? next_sym - first_sym : 0u;

  unsigned pos = s0, cnt = 0;
  bool live = pos < lim && want > 0;
  const unsigned cw0 = pos >> 6;
#pragma unroll
  for (int r = 0; r < kRingUnits; r++)  // initial fill of the ring
    ring[((cw0 + r) % kRingUnits) * 64 + lane] = load_unit(src, min(cw0 + r, nun));
  unsigned hi = cw0 + kRingUnits;
  unsigned long long win;
  unsigned avail, wnext;
  {
    const unsigned long long u0 = ring[(cw0 % kRingUnits) * 64 + lane];
    const unsigned long long u1 = ring[((cw0 + 1) % kRingUnits) * 64 + lane];
    const int sh = (int)(pos & 63);
    win = sh ? (u0 << sh) | (u1 >> (64 - sh)) : u0;  // 64 valid bits from pos on
    avail = 64;
    wnext = ((pos + 64) >> 5);
    const unsigned part = (pos + 64) & 31;  // (keep whole words only: refills stay word-aligned)
    avail -= part;
    win = part ? (win >> part) << part : win;
  }
  unsigned long long pf[kRingFetch];
  bool pending = false;
  const unsigned tsh = 32u - (unsigned)tb;
  unsigned short *my_stage = stage + lane * kStageStride;
  while (__any(live)) {
    unsigned got = 0;
    for (int rep = 0; rep < kRingBatch / kRingStep; rep++) {
      if (live && !pending) {  // request the next units; they are committed kRingStep symbols later
#pragma unroll
        for (int j = 0; j < kRingFetch; j++) pf[j] = load_unit(src, min(hi + j, nun));
        pending = true;
      }
#pragma unroll
      for (int k = 0; k < kRingStep; k++) {
        // top up 32 bits when 32 or fewer are left and the ring has the word (a lane out of units pauses)
        const unsigned un = wnext >> 1;
        const bool need = avail <= 32u && un < hi;
        const unsigned wd = ring32[((un % kRingUnits) * 64 + lane) * 2 + ((wnext & 1u) ^ 1u)];
        const unsigned long long add = (unsigned long long)wd << ((32u - avail) & 63u);
        win |= need ? add : 0ull;
        avail += need ? 32u : 0u;
        wnext += need ? 1u : 0u;
        const bool can = live && avail > 32u;
        unsigned e = table[(unsigned)(win >> 32) >> tsh];
        if (dbg & 2) e = (9u << 16) | 7u;  // (developer: every code 9 bits, no dependence on the table)
        if (__any(can && (int)e < 0)) {  // second level for some lane: the next sub_bits bits
          if (can && (int)e < 0) {
            const int sb = (int)((e >> 24) & 0x7f);
            e = table[(e & 0xffffff) + (unsigned)((win << tb) >> (64 - sb))];
          }
        }
        unsigned l = e >> 16, sym = e & 0xffffu;
        bool hit = e != 0;
        if (__any(can && !hit)) {  // a prefix the table leaves out: count the lengths (see k_decode_ring)
          if (can && !hit) {
            int ll = tb + 1;
            for (int j = tb + 1; j <= maxlen; j++) ll += win < slim[j] ? 1 : 0;
            if (ll <= maxlen) {
              const unsigned long long v = win >> (64 - ll);
              const unsigned long long kk = sentry[ll] + (v - sfirst[ll]);
              if (v >= sfirst[ll] && kk < (unsigned long long)dict) {
                sym = (unsigned)keys[kk] & 0xffffu;
                l = (unsigned)ll;
                hit = true;
              }
            }
          }
        }
        const bool good = can && hit && pos + l <= total;
        my_stage[got] = (unsigned short)sym;  // (counts only when the step was good: `got` moves on then)
        const unsigned lm = good ? l : 0u;
        pos = (can && !good) ? total : pos + lm;  // (a damaged stream ends the lane)
        win <<= lm;
        avail -= lm;
        got += good ? 1u : 0u;
        cnt += good ? 1u : 0u;
        live = live && (!can || (good && pos < lim && cnt < want));
      }
      // refill point: commit the units requested before these symbols if the ring has room
      if (pending && hi + kRingFetch <= (wnext >> 1) + kRingUnits) {
#pragma unroll
        for (int j = 0; j < kRingFetch; j++) ring[((hi + j) % kRingUnits) * 64 + lane] = pf[j];
        hi += kRingFetch;
        pending = false;
      }
    }
    if (!(dbg & 4)) {  // write-out: 64 / kRingBatch runs per instruction, the elements of a run side by side
      constexpr int RPI = 64 / kRingBatch;  // runs per store instruction
      const unsigned my_o = first_sym + cnt - got;
#pragma unroll 4
      for (int it = 0; it < 64 / RPI; it++) {
        const int L = it * RPI + lane / kRingBatch, k = lane % kRingBatch;
        const int n_L = __shfl((int)got, L, 64);
        const unsigned o_L = __shfl(my_o, L, 64);
        if (k < n_L && o_L + k < cap && !(dbg & 1)) dst[o_L + k] = (OUT)stage[L * kStageStride + k];
      }
    }
  }
}

} // namespace huff
} // namespace mgh
