// C ABI implementation (include/mgard_hip.h) of the MI355X-native MGARD-X hot
// path. Host orchestration only: the level loop of
// multi_dimension::decompose/recompose
// (reference include/mgard-x/DataRefactoring/MultiDimension/DataRefactoring.hpp:25-317)
// over the HIP kernels in kernels_*.hpp.
#include "../../include/mgard_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "env.hpp"
#include "hierarchy.hpp"
#include "kernels_v1.hpp"
#include "kernels_ipk.hpp"
#include "kernels_ipk_stream.hpp"
#include "kernels_ipk_spec.hpp"
#include "kernels_ipk_dma.hpp"
#include "kernels_fused.hpp"
#include "kernels_fused2.hpp"
#include "kernels_box.hpp"
#include "kernels_tail.hpp"
#include "kernels_recompose.hpp"
#include "kernels_recompose2.hpp"
#include "kernels_nd.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(_e == hipErrorOutOfMemory ? MGH_ERR_OUT_OF_MEMORY : MGH_ERR_DEVICE,          \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                          \
  } while (0)

struct ProfileEntry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0;
  uint64_t launches = 0;
};

} // namespace

struct mgh_hierarchy {
  int dtype = MGH_FLOAT;
  int device = 0;
  size_t num_cu = 256;  // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int D = 0, L = 0;
  uint64_t total = 0;
  uint64_t plane_elems = 0;  // product of the two fastest dimensions
  void *host = nullptr;  // HostHierarchy<T>*
  void *impl = nullptr;  // DeviceState<T>*
  bool profiling = false;
  int nd_rows = 1;  // MGH_ND_ROWS: the generic N-D path runs its row-wise kernels (kernels_nd.hpp; 0 = one thread per element, cross-check)
  bool force_nd_ipk = false;  // MGH_ND_IPK=1: the generic N-D path solves with its own one-thread-per-pencil kernel (cross-check)
  bool force_v1 = false;  // MGH_FORCE_V1=1: run the one-thread-per-element kernels only (unset: also for thin shapes, see mgh_hierarchy_create; 0: never)
  int force_v1_env = -1;
  bool force_nd = false;  // MGH_FORCE_ND=1: run the generic N-D kernels also for D <= 3 (cross-check)
  std::string prof_filter;  // empty = every kernel
  // MGH_IPK_STREAM: 1 = streaming Thomas solves (kernels_ipk_stream.hpp) on the levels whose
  // LDS-staged solve needs more than one round of resident workgroups (default), 0 = never
  int ipk_stream = 1;
  // MGH_IPK_DMA: 1 = strided float pencils whose tiles are all resident at once run k_ipk_dma
  // (kernels_ipk_dma.hpp: LDS-DMA front end, everything requested up front; default), 0 = never
  int ipk_dma = 1;
  long ipk_dma_min_env = -1;
  int inline_qp = 0;   // MGH_INLINE_QP: the finest level's kernel computes its quantizer itself (no k_make_qparams launch in front of it)
  int ipk_dma_rounds = 4;  // MGH_IPK_DMA_ROUNDS: k_ipk_dma also for levels whose tiles need up to this many rounds of resident workgroups
  size_t ipk_dma_min = 512;  // MGH_IPK_DMA_MIN: fewest tiles of a level for k_ipk_dma (two per CU; set in mgh_hierarchy_create)
  int absmax_warm_mb = 192;  // MGH_ABSMAX_WARM_MB: the norm pass reads all but the last so many MB of the input with nontemporal loads
  // MGH_FUSED_FACES: 1 = face tiles for the remainder columns / rows of a level (default), 0 = off
  int fused_faces = 1;
  int fused_tall = 1;  // MGH_FUSED_TALL: 64 x 4 tiles for levels with a short fastest extent (default 1)
  int slice_batch = 1;  // MGH_SLICE_BATCH: D = 4 decompression, all t-slices of a kind in one launch (default 1)
  int fused_xcd = 1;  // MGH_FUSED_XCD: tiles of a level in contiguous ranges per XCD (default 1; 2: the equal ranges of rounds 2-5 from 32 tiles on)
  int fused_fixed = 1; // MGH_FUSED_FIXED: the int64 + dictionary variant of the level kernel (default 1)
  int fused_wide = 1; // MGH_FUSED_WIDE: 4 x 64 tiles for 0 = no level, 1 = long marches, 2 = all (unset: 1 for floats, 0 for doubles)
  int fused4 = 1;     // MGH_FUSED4: D = 4 through the 3-D tile code, slice by slice (default 1)
  // MGH_BOX: levels up to this march class (0 = few tiles, 1 = mid-size, 2 = long marches) run
  // the box kernel (kernels_box.hpp: no march, every phase once over a 4 x 4 x 8 box) instead of
  // the marching tile kernel; 0 = none, 1 = class 0 (default), 2 = classes 0-1, 3 = every level
  int box = 1;
  int ipk_spec = 1;     // MGH_IPK_SPEC: few long contiguous pencils (1-D arrays) are solved in chunks, each verified against the sequential sweep (kernels_ipk_spec.hpp); 0 = one lane per pencil
  int outlier_agg = 2;  // MGH_OUTLIER_AGG: the level kernel asks for outlier slots once per workgroup and pair step instead of once per wave and plane (kernels_fused2.hpp: OutlierShared): 0 never, 1 always, 2 when the previous call on this hierarchy left more than 0.5 % of its values (and more than 200 000) in the outlier list
  int ipk_chunk = 1;    // MGH_IPK_CHUNK: the LDS-staged solve of contiguous pencils shares a tile's sweeps between the four waves (thomas_chunked: chunks verified against the sequential sweep)
  int ipk_chunk_k = 0;  // MGH_IPK_CHUNK_K: warm-up length of a chunk (0 = from the tables, chunk_warmup_need; small values make the verification fail and exercise the fall-back)
  int ipk_chunk_need = 0;  // warm-up length that the Thomas tables of this hierarchy need (set with the tables)
  int ipk_spec_long = 1024;  // MGH_IPK_SPEC_LONG: strided pencils of this length and more, one round of tiles at most, run in verified chunks too (0: never)
  uint32_t ipk_spec_max = 16384;  // MGH_IPK_SPEC_MAX: most pencils of a solve whose pencils do not fit LDS that still run in verified chunks
  int ipk_spec_k = 0;   // MGH_IPK_SPEC_K: warm-up length of a chunk (0 = 64 floats / 128 doubles; tiny values make the verification fail and exercise the repair)
  int sym16_mixed = 1;  // MGH_SYM16_MIXED: 16-bit symbols for the finest level only, int64 below it (default), 0 = 16-bit symbols on every level
  int restore_v = 3;  // MGH_RESTORE_V: 3 = marching node restore (kernels_recompose2.hpp), 2 = one wave per pair of fine rows
  int tail_solves = 1;  // MGH_TAIL_SOLVES: the tail kernel runs the Thomas solves of the level above it
  // (the rest of the developer switches, env.hpp; all read when the hierarchy is created)
  size_t cls1 = 256, cls2 = 2048;  // MGH_CLS1 / MGH_CLS2: tile-count thresholds of the march classes
  int rch[3] = {1, 4, 16};         // MGH_RCH=a,b,c: coarse planes per workgroup of the three classes
  uint32_t ipk_w = 64;             // MGH_IPK_W: widest solver wave of the streaming Thomas solves
  int ipk_pd = 1;                  // MGH_IPK_PD: their load-pipeline depth
  size_t ipk_contig_rounds = 4;    // MGH_IPK_CONTIG: rounds of the LDS-staged contiguous solve from which the streaming one takes over
  int ipk_range_mb = 128;          // MGH_IPK_RANGE_MB: f- and c-solve of a load vector bigger than twice this run in r-plane ranges of this size (0 = off)
  int ipk_kr16 = 1;                // MGH_IPK_KR16: 16 register-resident batches for float pencils of 512+ elements
  size_t ipk_wpc = 8;              // MGH_IPK_WPC: most one-wave solver workgroups per CU the host plans with
  bool no_head = false;            // MGH_NO_RECOMPOSE_HEAD
  bool restore_rows = false;       // MGH_RESTORE_ROWS
  bool debug_sync = false;         // MGH_DEBUG_SYNC: name every launch on stderr and synchronise behind it
  std::map<std::string, ProfileEntry> prof;
  size_t device_bytes = 0;
  uint64_t shape[MGH_MAX_DIM] = {};
  // mgh_set_ld: leading dimensions of the caller's T arrays, [MGH_LD_IN / MGH_LD_OUT][dim]
  uint64_t ld[2][MGH_MAX_DIM] = {};
  bool has_ld[2] = {false, false};
  bool ld_guard = false;  // inside an entry point that has already dealt with the leading dimensions
};

namespace {

using namespace mgh;

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};

// Warm-up length of the chunked sweeps (thomas_chunked) for one Thomas table: a chunk starts
// from state 0 instead of the true state, an error of the size of the data; after K steps it is
// that times the product of the K multipliers it passed. For the two runs to MEET (and not
// only to be close) the error has to fall far below one unit in the last place: the smallest K
// for which every window of K consecutive multipliers has a product below 2^-(mantissa + 12),
// rounded up to 8. (A bad guess costs time, not correctness: a chunk that has not met is
// detected and the tile solved again by one lane per pencil.)
template <typename T> int chunk_warmup_need(const std::vector<T> &tt) {
  const size_t n = tt.size() / 3;
  if (n < 2) return 0;
  const double target = -((sizeof(T) == 4 ? 24 : 53) + 12);
  std::vector<double> lf(n), lb(n);
  for (size_t i = 0; i < n; i++) {
    const double f = std::fabs((double)tt[i]);
    const double b = tt[2 * n + i] != 0 ? std::fabs((double)tt[n + i] / (double)tt[2 * n + i]) : 0.0;
    lf[i] = f > 0 ? std::log2(f) : -64.0;
    lb[i] = b > 0 ? std::log2(b) : -64.0;
  }
  int need = 0;
  for (const auto *lg : {&lf, &lb}) {
    for (size_t i = 0; i < n; i++) {  // windows starting at i (either direction: the same products)
      double acc = 0;
      size_t k = 0;
      while (i + k < n && acc > target) acc += (*lg)[i + k++];
      if (acc > target) break;  // the window ran into the end of the table: shorter ones are exact starts
      need = std::max(need, (int)k);
    }
  }
  return (need + 7) / 8 * 8;
}

template <typename T> struct LevelTables {
  // index k = 0,1,2 <-> (r, c, f) of the 3-D view; nullptr for inactive dims
  const T *ratio[3] = {nullptr, nullptr, nullptr};   // fine level l
  const T *mass[3] = {nullptr, nullptr, nullptr};    // l -> l-1
  const T *thomas[3] = {nullptr, nullptr, nullptr};  // coarse level l-1
  Box3 box;
  bool active[3] = {false, false, false};
};

template <typename T> struct DeviceState {
  HostHierarchy<T> *hh = nullptr;
  T *tables = nullptr;
  int *marks = nullptr;
  std::vector<LevelTables<T>> lt;  // [l], l >= 1 (3-D view, D <= 3)
  std::vector<size_t> lt_end;      // [l]: element offset in `tables` where the tables of level l end
                                   // (levels are laid out 1, 2, ... back to back: the tail kernel
                                   // copies the block of its levels to LDS in one pass)
  struct NdLevel {
    const T *ratio[kNd], *mass[kNd], *thomas[kNd];
  };
  std::vector<NdLevel> nd;         // [l], l >= 1 (all D dims; used by the D > 3 path)
  T *nd_w = nullptr, *nd_a = nullptr, *nd_b = nullptr;  // N-D scratch (lazily allocated)
  // fused D = 4 path (lazily allocated): compact coarse arrays per level, per-slice load vectors
  // of the biggest level (padded slice positions), t-swept load vector / correction
  std::vector<T *> nodal4;
  T *load4 = nullptr, *corr4 = nullptr;
  bool state4_ready = false;       // set only once every allocation of ensure_state4 succeeded
  std::vector<T *> nodal;          // [l] compact nodal buffers, l = 0..L-1
  T *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
  T *scratch_full = nullptr;       // lazily allocated full-size copy
  T *qz = nullptr;                 // 2*(L+1): quantizers, volumes
  unsigned long long *scalar = nullptr;  // 8-byte device scalar (norm / counters)
  // fused device-norm path: fscal[slot] is zero on entry, the other slot is zeroed by
  // k_make_qparams for the next call; `fscal_dirty` marks a call that died in between
  unsigned long long *fscal = nullptr;
  int scalar_slot = 0;
  bool fscal_dirty = false;
  // mgh_norm_stream_begin/add: the slot of the NEXT fused call already holds the reduction of its
  // input (accumulated slab by slab while the input was arriving from the host)
  bool norm_streamed = false;
  // MGH_INLINE_QP: quantizer constants of the last inline call, on the device and as uploaded
  QParamArgs<T> *qinl_dev = nullptr;
  QParamArgs<T> qinl_host;
  bool qinl_valid = false;
  T *normval = nullptr;                  // norm as T, written by k_make_qparams
  unsigned long long *oh_key = nullptr;  // outlier table of the 16-bit symbol path (grown on demand)
  long long *oh_val = nullptr;
  size_t oh_slots = 0;
  int64_t *qbox = nullptr;               // ... and its compact int64 copy of the coarse corner box
  size_t qbox_elems = 0;
  // chunked Thomas solves of few long pencils (kernels_ipk_spec.hpp): forward results, chunk-edge values
  T *spec_y = nullptr, *spec_a = nullptr, *spec_b = nullptr;
  size_t spec_y_elems = 0, spec_edge_elems = 0;
  unsigned long long *spec_fixed = nullptr;  // chunks that had to be recomputed (diagnostics)
  // outlier count of an earlier call, written by its last kernel into host memory (hipHostMalloc;
  // the device writes through the same pointer): picks the level kernel's variant, see outlier_agg
  unsigned long long *outliers_seen = nullptr;
  QuantMeta qmeta;
  size_t full_I = 0, full_J = 0;   // strides of the full array in the 3-D view
  // Strides of a PITCHED caller array for the one call that set them (mgh_set_ld; 0 = full_I /
  // full_J): the finest level's input of the fused decomposition, the finest level's output of the
  // fused recomposition. Every other path sees dense copies (ld_pack / ld_unpack).
  size_t src_I = 0, src_J = 0, dst_I = 0, dst_J = 0;
  T *pack_in = nullptr, *pack_out = nullptr;  // dense copies of pitched arrays (lazily allocated)
};

template <typename T> HostHierarchy<T> *HH(const mgh_hierarchy *h) {
  return static_cast<HostHierarchy<T> *>(h->host);
}
template <typename T> DeviceState<T> *DS(const mgh_hierarchy *h) {
  return static_cast<DeviceState<T> *>(h->impl);
}

// ---- profiled launch -------------------------------------------------------
template <typename F> int launch(mgh_hierarchy *h, const char *name, hipStream_t s, F &&f) {
  // MGH_DEBUG_SYNC=1: name every launch on stderr and synchronise behind it, so that a GPU
  // memory fault can be attributed to a kernel (developer aid)
  if (h->debug_sync) {
    std::fprintf(stderr, "[mgh] %s\n", name);
    std::fflush(stderr);
    f();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return MGH_SUCCESS;
  }
  if (!h->profiling || (!h->prof_filter.empty() && h->prof_filter != name)) {
    f();
    HIP_TRY(hipGetLastError());
    return MGH_SUCCESS;
  }
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  HIP_TRY(hipEventRecord(a, s));
  f();
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b, s));
  h->prof[name].pending.emplace_back(a, b);
  return MGH_SUCCESS;
}

// The same for ONE kernel whose timing is read inside timed regions (the level pass: bench.py
// keeps HIP events on the dominant kernel during the timed steps): hipExtLaunchKernelGGL stamps
// the start and stop events from the dispatch itself instead of two marker packets around the
// launch. Cost of the events per 512^3 step (tools/exp_gap.py, 20 steps each, same box): none
// 0.899-0.905 ms, marker packets +4 us, kernel-attached +1.5 us. (The ~6 us gaps a rocprofv3
// timeline shows on either side of the profiled kernel are there with both kinds of events.)
template <typename K, typename... Args>
int launch_kernel(mgh_hierarchy *h, const char *name, hipStream_t s, K kernel, dim3 grid, dim3 block,
                  size_t lds, Args... args) {
  if (h->debug_sync || !h->profiling || (!h->prof_filter.empty() && h->prof_filter != name))
    return launch(h, name, s, [&] { hipLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, args...); });
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, a, b, 0, args...);
  HIP_TRY(hipGetLastError());
  h->prof[name].pending.emplace_back(a, b);
  return MGH_SUCCESS;
}

#define TRY(expr)                 \
  do {                            \
    int _rc = (expr);             \
    if (_rc != MGH_SUCCESS) return _rc; \
  } while (0)

inline dim3 grid3(uint32_t n0, uint32_t n1, uint32_t n2, dim3 blk) {
  return dim3((n2 + blk.x - 1) / blk.x, (n1 + blk.y - 1) / blk.y, n0);
}

template <typename T> int dev_alloc(mgh_hierarchy *h, T **p, size_t count) {
  *p = nullptr;
  if (count == 0) return MGH_SUCCESS;
  HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
  h->device_bytes += count * sizeof(T);
  return MGH_SUCCESS;
}

// mirror of dev_alloc: frees *p (if any) and takes its bytes out of the handle's footprint
template <typename T> void dev_free(mgh_hierarchy *h, T **p, size_t count) {
  if (!*p) return;
  if (hipFree(*p) == hipSuccess) h->device_bytes -= std::min(h->device_bytes, count * sizeof(T));
  *p = nullptr;
}

template <typename T> int build_device_state(mgh_hierarchy *h) {
  HostHierarchy<T> *hh = HH<T>(h);
  auto *ds = new DeviceState<T>();
  h->impl = ds;
  ds->hh = hh;
  const int D = hh->D, L = hh->L;
  if (D > 3) {
    // generic per-dim tables for the N-D kernels
    std::vector<T> arena;
    auto push = [&](const std::vector<T> &v) {
      size_t off = arena.size();
      arena.insert(arena.end(), v.begin(), v.end());
      while (arena.size() % 4) arena.push_back(0);
      return off;
    };
    struct Off { size_t r[kNd], m[kNd], t[kNd]; };
    std::vector<Off> offs(L + 1);
    for (int l = 1; l <= L; l++)
      for (int d = 0; d < D; d++) {
        offs[l].r[d] = push(hh->lv[l][d].ratio);
        offs[l].m[d] = push(hh->mass_table(l, d));
        offs[l].t[d] = push(hh->thomas_table(l - 1, d));
      }
    TRY(dev_alloc(h, &ds->tables, arena.size()));
    HIP_TRY(hipMemcpy(ds->tables, arena.data(), arena.size() * sizeof(T), hipMemcpyHostToDevice));
    ds->nd.resize(L + 1);
    for (int l = 1; l <= L; l++)
      for (int d = 0; d < D; d++) {
        ds->nd[l].ratio[d] = ds->tables + offs[l].r[d];
        ds->nd[l].mass[d] = ds->tables + offs[l].m[d];
        ds->nd[l].thomas[d] = ds->tables + offs[l].t[d];
      }
    std::vector<int> marks;
    ds->qmeta.D = D;
    ds->qmeta.calc_vol = 0;
    for (int d = 0; d < D; d++) {
      ds->qmeta.shape[d] = (uint32_t)hh->shape[d];
      ds->qmeta.markoff[d] = (uint32_t)marks.size();
      marks.insert(marks.end(), hh->marks[d].begin(), hh->marks[d].end());
    }
    TRY(dev_alloc(h, &ds->marks, marks.size()));
    HIP_TRY(hipMemcpy(ds->marks, marks.data(), marks.size() * sizeof(int), hipMemcpyHostToDevice));
    TRY(dev_alloc(h, &ds->qz, (size_t)2 * (L + 1)));
    TRY(dev_alloc(h, &ds->scalar, (size_t)2));
    TRY(dev_alloc(h, &ds->fscal, (size_t)2));
    HIP_TRY(hipMemset(ds->fscal, 0, 16));
    TRY(dev_alloc(h, &ds->normval, (size_t)2));
    return MGH_SUCCESS;
  }

  // ---- spacing tables: one arena, one upload --------------------------------
  std::vector<T> arena;
  auto push = [&](const std::vector<T> &v) {
    size_t off = arena.size();
    arena.insert(arena.end(), v.begin(), v.end());
    while (arena.size() % 4) arena.push_back(0);  // keep 16-byte alignment for f32
    return off;
  };
  struct Off {
    size_t ratio[3], mass[3], thomas[3];
  };
  std::vector<Off> offs(L + 1);
  ds->lt.resize(L + 1);
  for (int l = 1; l <= L; l++) {
    LevelTables<T> &t = ds->lt[l];
    for (int k = 0; k < 3; k++) {
      const int d = D - 3 + k;
      if (d < 0) {
        t.box.n[k] = t.box.m[k] = 1;
        t.active[k] = false;
        continue;
      }
      t.active[k] = true;
      t.box.n[k] = (uint32_t)hh->level_shape[l][d];
      t.box.m[k] = (uint32_t)hh->level_shape[l - 1][d];
      offs[l].ratio[k] = push(hh->lv[l][d].ratio);
      offs[l].mass[k] = push(hh->mass_table(l, d));
      const std::vector<T> tt = hh->thomas_table(l - 1, d);
      h->ipk_chunk_need = std::max(h->ipk_chunk_need, chunk_warmup_need(tt));
      offs[l].thomas[k] = push(tt);
    }
  }
  TRY(dev_alloc(h, &ds->tables, arena.size()));
  HIP_TRY(hipMemcpy(ds->tables, arena.data(), arena.size() * sizeof(T), hipMemcpyHostToDevice));
  ds->lt_end.assign(L + 1, 0);
  for (int l = 1; l <= L; l++) {
    size_t next = arena.size();
    for (int l2 = l + 1; l2 <= L && next == arena.size(); l2++)
      for (int k = 0; k < 3; k++)
        if (ds->lt[l2].active[k]) {
          next = offs[l2].ratio[k];
          break;
        }
    ds->lt_end[l] = next;
  }
  for (int l = 1; l <= L; l++)
    for (int k = 0; k < 3; k++)
      if (ds->lt[l].active[k]) {
        ds->lt[l].ratio[k] = ds->tables + offs[l].ratio[k];
        ds->lt[l].mass[k] = ds->tables + offs[l].mass[k];
        ds->lt[l].thomas[k] = ds->tables + offs[l].thomas[k];
      }

  // per-dim view of the same tables for the generic N-D kernels (cross-check path)
  ds->nd.resize(L + 1);
  for (int l = 1; l <= L; l++)
    for (int d = 0; d < D; d++) {
      const int k = d + 3 - D;
      ds->nd[l].ratio[d] = ds->lt[l].ratio[k];
      ds->nd[l].mass[d] = ds->lt[l].mass[k];
      ds->nd[l].thomas[d] = ds->lt[l].thomas[k];
    }

  // ---- level marks ------------------------------------------------------------
  std::vector<int> marks;
  ds->qmeta.D = D;
  ds->qmeta.calc_vol = 0;
  for (int d = 0; d < D; d++) {
    ds->qmeta.shape[d] = (uint32_t)hh->shape[d];
    ds->qmeta.markoff[d] = (uint32_t)marks.size();
    marks.insert(marks.end(), hh->marks[d].begin(), hh->marks[d].end());
  }
  TRY(dev_alloc(h, &ds->marks, marks.size()));
  HIP_TRY(hipMemcpy(ds->marks, marks.data(), marks.size() * sizeof(int), hipMemcpyHostToDevice));

  // ---- workspace ----------------------------------------------------------------
  ds->nodal.assign(L + 1, nullptr);
  for (int l = 0; l < L; l++) {
    size_t cnt = 1;
    for (int d = 0; d < D; d++) cnt *= hh->level_shape[l][d];
    TRY(dev_alloc(h, &ds->nodal[l], cnt));
  }
  if (L >= 1) {
    const Box3 &b = ds->lt[L].box;
    // t1 / t2 (intermediate sweeps of the one-thread-per-element path) are allocated on first
    // use: the fused 3-D path never needs them
    TRY(dev_alloc(h, &ds->t3, (size_t)b.m[0] * b.m[1] * b.m[2]));
  }
  TRY(dev_alloc(h, &ds->qz, (size_t)2 * (L + 1)));
  TRY(dev_alloc(h, &ds->scalar, (size_t)2));
  TRY(dev_alloc(h, &ds->fscal, (size_t)2));
  HIP_TRY(hipMemset(ds->fscal, 0, 16));
  TRY(dev_alloc(h, &ds->normval, (size_t)2));
  ds->full_J = hh->shape[D - 1];
  ds->full_I = (D >= 2 ? hh->shape[D - 2] : 1) * ds->full_J;
  return MGH_SUCCESS;
}

template <typename T> void destroy_state(mgh_hierarchy *h) {
  auto *ds = DS<T>(h);
  if (ds) {
    (void)hipFree(ds->tables);
    (void)hipFree(ds->marks);
    for (T *p : ds->nodal) (void)hipFree(p);
    (void)hipFree(ds->t1);
    (void)hipFree(ds->t2);
    (void)hipFree(ds->t3);
    (void)hipFree(ds->scratch_full);
    (void)hipFree(ds->qinl_dev);
    (void)hipFree(ds->pack_in);
    (void)hipFree(ds->pack_out);
    (void)hipFree(ds->nd_w);
    (void)hipFree(ds->nd_a);
    (void)hipFree(ds->nd_b);
    for (T *p : ds->nodal4) (void)hipFree(p);
    (void)hipFree(ds->load4);
    (void)hipFree(ds->corr4);
    (void)hipFree(ds->qz);
    (void)hipFree(ds->scalar);
    (void)hipFree(ds->fscal);
    (void)hipFree(ds->normval);
    (void)hipFree(ds->oh_key);
    (void)hipFree(ds->oh_val);
    (void)hipFree(ds->qbox);
    (void)hipFree(ds->spec_y);
    (void)hipFree(ds->spec_a);
    (void)hipFree(ds->spec_b);
    (void)hipFree(ds->spec_fixed);
    (void)hipHostFree(ds->outliers_seen);
    delete ds;
  }
  delete HH<T>(h);
}

template <typename T> int ensure_scratch(mgh_hierarchy *h) {
  auto *ds = DS<T>(h);
  if (!ds->scratch_full) TRY(dev_alloc(h, &ds->scratch_full, (size_t)h->total));
  return MGH_SUCCESS;
}

// LDS budget for the IPK tiles: whole pencils of 64 (or 32) lanes must fit.
constexpr size_t kLdsPerCU = 160 * 1024;

// (`bytes`: the dynamic part; a kernel with static LDS of its own asks for that much less)
template <typename K> int allow_big_lds(K kernel, size_t bytes = kLdsPerCU) {
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return MGH_SUCCESS;
}
// ... once per DEVICE (the attribute belongs to the function on the current device) and safe
// from several host threads: `done` holds one bit per device ordinal.
template <typename K> int allow_big_lds_once(K kernel, std::atomic<uint64_t> &done, size_t bytes = kLdsPerCU) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const uint64_t bit = (uint64_t)1 << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return MGH_SUCCESS;
  TRY(allow_big_lds(kernel, bytes));
  done.fetch_or(bit, std::memory_order_release);
  return MGH_SUCCESS;
}

// Thomas solve along `axis` of the compact (m[0], m[1], m[2]) box.
// axis 0 only: `nbatch` boxes `batch_stride` elements apart in ONE launch (the slices of a 4-D level).
template <typename T>
int ipk_launch(mgh_hierarchy *h, int axis, const uint32_t *m, T *x, const T *tt, T *add_to,
               int sign, hipStream_t s, uint32_t nbatch = 1, size_t batch_stride = 0) {
  if (nbatch > 1 && axis != 0) return fail(MGH_ERR_INVALID_ARGUMENT, "ipk_launch: batches along the slowest axis only");
  const uint32_t n = m[axis];
  static const char *names[3] = {"ipk_r", "ipk_c", "ipk_f"};
  const char *name = names[axis];
  // tile width (pencils per workgroup): whole pencils must fit in LDS; among the fitting
  // widths take the one that needs the fewest "rounds" of resident workgroups
  const size_t pencil_bytes = (size_t)(n + (axis == 2 && n % 2 == 0 ? 1 : 0)) * sizeof(T);
  const uint32_t npencil = axis == 2 ? m[0] * m[1] : (axis == 1 ? m[0] * m[2] : nbatch * m[1] * m[2]);
  // Few long pencils (a 1-D array: ONE pencil per level): parallel inside the pencil, every chunk
  // verified against the sequential sweep (kernels_ipk_spec.hpp)
  auto spec_solve = [&]() -> int {
    auto *ds = DS<T>(h);
    SpecGeom G{};
    G.n = n;
    G.K = h->ipk_spec_k > 0 ? (uint32_t)h->ipk_spec_k : (sizeof(T) == 4 ? 64u : 128u);
    uint32_t S = std::max<uint32_t>(128, std::min<uint32_t>(1024, n / 16384));
    G.S = (S + 7) / 8 * 8;
    G.nchunk = (n + G.S - 1) / G.S;
    G.npencil = npencil;
    if (axis == 2) { G.n_inner = npencil; G.outer_stride = 0; G.inner_stride = n; G.stride = 1; G.along_p = 0; }
    else if (axis == 1) { G.n_inner = m[2]; G.outer_stride = (size_t)m[1] * m[2]; G.inner_stride = 1; G.stride = m[2]; G.along_p = 1; }
    else { G.n_inner = m[1] * m[2]; G.outer_stride = batch_stride; G.inner_stride = 1; G.stride = (size_t)m[1] * m[2]; G.along_p = 1; }  // (batches: boxes back to back, spec_ok)
    const uint32_t nchunk = G.nchunk;
    const size_t total = (size_t)npencil * n, edges = (size_t)npencil * nchunk;
    if (total > ds->spec_y_elems) {
      (void)hipFree(ds->spec_y);
      ds->spec_y = nullptr;
      ds->spec_y_elems = 0;
      HIP_TRY(hipMalloc(&ds->spec_y, total * sizeof(T)));
      ds->spec_y_elems = total;
    }
    if (edges > ds->spec_edge_elems) {
      (void)hipFree(ds->spec_a);
      (void)hipFree(ds->spec_b);
      ds->spec_a = ds->spec_b = nullptr;
      ds->spec_edge_elems = 0;
      HIP_TRY(hipMalloc(&ds->spec_a, edges * sizeof(T)));
      HIP_TRY(hipMalloc(&ds->spec_b, edges * sizeof(T)));
      ds->spec_edge_elems = edges;
    }
    if (!ds->spec_fixed) {  // [0]: chunks repaired so far, [1]: mismatch flags of the two sweeps of a call
      HIP_TRY(hipMalloc(&ds->spec_fixed, 16));
      HIP_TRY(hipMemsetAsync(ds->spec_fixed, 0, 16, s));
    }
    const unsigned grid = (unsigned)((edges + 63) / 64);
    const unsigned pgrid = (npencil + 63) / 64;
    T *y = ds->spec_y, *ea = ds->spec_a, *eb = ds->spec_b;
    unsigned long long *fx = ds->spec_fixed;
    unsigned *mm = reinterpret_cast<unsigned *>(ds->spec_fixed + 1);
    const unsigned cgrid = (unsigned)((edges + 255) / 256);
    HIP_TRY(hipMemsetAsync(mm, 0, 8, s));
    TRY(launch(h, name, s, [&] {
      k_ipk_spec_fwd<T><<<grid, 64, 0, s>>>(G, x, y, tt, ea, eb);
      k_ipk_spec_check<T><<<cgrid, 256, 0, s>>>(nchunk, npencil, ea, eb, +1, mm);
      k_ipk_spec_fix<T><<<pgrid, 64, 0, s>>>(G, x, y, tt, ea, eb, +1, fx, mm);
      k_ipk_spec_bwd<T><<<grid, 64, 0, s>>>(G, y, x, tt, ea, eb);
      k_ipk_spec_check<T><<<cgrid, 256, 0, s>>>(nchunk, npencil, ea, eb, -1, mm + 1);
      k_ipk_spec_fix<T><<<pgrid, 64, 0, s>>>(G, y, x, tt, ea, eb, -1, fx, mm + 1);
      if (add_to)
        k_ipk_spec_apply<T><<<(unsigned)std::min<size_t>((total + 255) / 256, 4096), 256, 0, s>>>(total, add_to, x, sign);
    }));
    return MGH_SUCCESS;
  };
  // (the batches of a 4-D level: only back to back -- the chunk buffers and the add-to pass see one array)
  const bool spec_ok = h->ipk_spec && (nbatch == 1 || batch_stride == (size_t)m[0] * m[1] * m[2]);
  if (axis == 2 && nbatch == 1 && h->ipk_spec && npencil <= 64 && n >= 2048) return spec_solve();
  int best_w = 0;
  size_t best_rounds = ~(size_t)0;
  for (int w : {64, 48, 32, 16}) {
    const size_t lds = w * pencil_bytes;
    if (lds > kLdsPerCU) continue;
    const size_t per_cu = std::min<size_t>(kLdsPerCU / lds, 8);
    const size_t blocks = (npencil + w - 1) / w;
    const size_t rounds = (blocks + per_cu * h->num_cu - 1) / (per_cu * h->num_cu);
    if (rounds < best_rounds) {
      best_rounds = rounds;
      best_w = w;
    }
  }
  // Long strided pencils, few enough of them to be one round of tiles with most of the chip idle
  // (16395 x 64 x 64: the r-solve of the 2051 x 9 x 9 level is 6 tiles and a chain of 2051 steps down
  // and 2051 back at ~40 ns each in LDS -- 161 us for 0.7 MB; the levels above it 91 and 53 us): the
  // verified chunks put a wave on every piece of every pencil. MGH_IPK_SPEC_LONG: the pencil length
  // from which on (default 1024; 0: never).
  if (axis != 2 && spec_ok && h->ipk_spec_long && n >= (uint32_t)h->ipk_spec_long &&
      npencil <= 64u * (uint32_t)h->num_cu && npencil <= h->ipk_spec_max)
    return spec_solve();
  // Contiguous pencils, LDS-staged tiles whose sweeps are shared by the four waves (thomas_chunked;
  // warm-up length from the tables, ipk_chunk_need), ahead of the streaming kernels: 512^3 f32 top
  // level 53 -> 44 us, f64 120 -> 109 us, 1024^3 554 -> 543 us. (The same for strided pencils and
  // for the plane kernel of the small levels was measured and dropped: profiles/NOTES.md.)
  if (h->ipk_chunk && axis == 2 && best_w && n >= 64 && best_w * pencil_bytes + 8192 <= kLdsPerCU) {
    const uint32_t K = h->ipk_chunk_k > 0 ? (uint32_t)h->ipk_chunk_k : (uint32_t)h->ipk_chunk_need;
    if (K > 0 && K <= n / 2) {
      const uint32_t pad = (n % 2 == 0) ? 1u : 0u;
      const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + n - 1) / n);
      const uint32_t P = (uint32_t)best_w;
      static std::atomic<uint64_t> once{0};
      TRY(allow_big_lds_once(k_ipk_lds_contig<T, true>, once, kLdsPerCU - 8192));
      return launch(h, name, s, [&] {
        k_ipk_lds_contig<T, true><<<(npencil + P - 1) / P, 256, P * pencil_bytes, s>>>(
            npencil, n, pad, magic, P, x, tt, add_to, sign, K);
      });
    }
  }
  // Streaming solves: every wave a solver, forward results parked in registers + LDS +
  // (the leading n_glob elements) in place in global memory. Where the forward results are parked
  // decides how many pencils a CU works on at once, and residency is the throughput of these
  // latency-bound chains: the host picks the tile width W and n_glob that need the FEWEST rounds
  // of resident workgroups (up to ipk_wpc one-wave workgroups per CU), then the least global
  // parking. Used when the LDS-staged tiles need more than one round (strided pencils), or --
  // contiguous pencils, where the LDS-staged kernel is the better one at one or two rounds --
  // from four rounds on (1024^3: 2 KB pencils leave ONE staged tile per CU, 16 rounds).
  // batches of 64 bytes per lane: 16 floats / 8 doubles; the last KR batches stay in registers.
  // KR = 8 (a third of the register file: two or more waves per SIMD), or -- float pencils of
  // 512+ elements, MGH_IPK_KR16 -- KR = 16: 256 values of every pencil in registers, one wave per
  // SIMD, so that most of the rest fits in LDS and little is parked in global memory (1024^3:
  // PMC traffic of a solve 2.0-2.5 GB for 1.08 GB algorithmic with KR = 8).
  constexpr int kNotApplicable = 1 << 20;
  constexpr uint32_t U = sizeof(T) == 4 ? 16 : 8;
  // Strided float pencils, every tile of the level resident at once: the LDS-DMA variant
  // (kernels_ipk_dma.hpp). KR register-resident batches: as many as keep two waves per SIMD.
  if constexpr (sizeof(T) == 4) {
    const size_t box_bytes = (nbatch > 1 ? nbatch * batch_stride : (size_t)m[0] * m[1] * m[2]) * sizeof(T);
    if (h->ipk_dma && axis != 2 && box_bytes < ((size_t)1 << 32)) {
      auto try_kr = [&](auto KRc) -> int {
        constexpr uint32_t KR = decltype(KRc)::value;
        if (n < KR * U) return kNotApplicable;
        const uint32_t nl = n - KR * U;
        const size_t lds = (size_t)nl * 64 * sizeof(T);
        if (lds > kLdsPerCU - 4096) return kNotApplicable;
        const size_t per_cu = lds ? std::min<size_t>((kLdsPerCU - 4096) / lds, 8) : 8;
        const size_t tiles = ((size_t)npencil + 63) / 64;
        // (a level with fewer than two tiles per CU is served better by the LDS-staged kernels, whose
        // four waves per tile stream it in and out: 129^3, 261 tiles, 18.8 vs 16.1 us)
        // (MGH_IPK_DMA_MIN: that threshold in tiles, 0 in the tests that run this kernel on small shapes)
        // (round 6: short pencils -- four or more tiles per CU -- also when the level needs up to
        // MGH_IPK_DMA_ROUNDS rounds of resident workgroups: the 8 x 512^3 slab's 5168 tiles of
        // 257-element pencils, ipk_c 242 -> 213 us, ipk_r 204 -> 190 us against k_ipk_stream; long
        // pencils with ONE tile per CU lose badly that way -- 1024^3's 513-element pencils 430 -> 850 us)
        const size_t rounds = per_cu >= 4 ? (size_t)h->ipk_dma_rounds : 1;
        if (tiles > per_cu * h->num_cu * rounds || tiles < h->ipk_dma_min) return kNotApplicable;
        const unsigned blocks = (unsigned)((tiles + 7) / 8 * 8);
        uint32_t n_inner;
        size_t outer_stride, stride;
        if (axis == 1) { n_inner = m[2]; outer_stride = (size_t)m[1] * m[2]; stride = m[2]; }
        else { n_inner = m[1] * m[2]; outer_stride = batch_stride; stride = (size_t)m[1] * m[2]; }
#define MGH_DMA(ADD)                                                                          \
  {                                                                                           \
    static std::atomic<uint64_t> once{0};                                                     \
    TRY(allow_big_lds_once(k_ipk_dma<T, U, KR, ADD>, once));                                  \
    return launch(h, name, s, [&] {                                                           \
      k_ipk_dma<T, U, KR, ADD><<<blocks, 64, lds, s>>>(npencil, n_inner, outer_stride, 1,     \
                                                       stride, n, x, tt, add_to);             \
    });                                                                                       \
  }
        if (!add_to) MGH_DMA(0)
        if (sign > 0) MGH_DMA(1)
        MGH_DMA(-1)
#undef MGH_DMA
      };
      const int rc = try_kr(std::integral_constant<uint32_t, 10>{});
      if (rc != kNotApplicable) return rc;
    }
  }
  auto stream_plan = [&](auto KRc, size_t wpc_cap) -> int {
    constexpr uint32_t KR = decltype(KRc)::value;
    const size_t wpc = std::min(h->ipk_wpc, wpc_cap);
    const uint32_t nb = n / U;
    const size_t box_bytes = (nbatch > 1 ? nbatch * batch_stride : (size_t)m[0] * m[1] * m[2]) * sizeof(T);
    // (strided pencils: measured inside the step at 512^3, ipk_c 59 -> 52 us, ipk_r of
    // all levels 128 -> 103 us, but the contiguous solve 60 -> 65 us)
    const size_t min_rounds = axis == 2 ? h->ipk_contig_rounds : 2;
    if (h->ipk_stream && best_w && best_rounds >= min_rounds && nb >= KR &&
        box_bytes < ((size_t)1 << 32)) {
      const uint32_t parked = (nb - KR) * U;  // elements per pencil outside the registers
      uint32_t W = 0, n_glob = 0;
      size_t w_rounds = ~(size_t)0;
      const uint32_t w_max = h->ipk_w;
      for (uint32_t ng = 0; ng <= parked; ng += U) {
        for (uint32_t w : {64u, 60u, 56u, 48u, 40u, 32u, 24u, 16u}) {
          if (w > w_max) continue;
          const size_t lds = (size_t)w * (parked - ng) * sizeof(T) +
                             (axis == 2 ? TileIO<T, U>::stage_elems * sizeof(T) : 0);
          // ~230 VGPRs (KR = 8): two waves per SIMD = 8 one-wave workgroups per CU; KR = 16:
          // ~400 VGPRs, one wave per SIMD = 4 per CU (the caps the host plans with)
          // (five 32 KB allocations do not fit one CU although 5 * 32 KB = 160 KB: leave a margin)
          const size_t per_cu = lds ? std::min<size_t>((kLdsPerCU - 4096) / lds, wpc) : wpc;
          if (!per_cu) continue;
          const size_t blocks = ((size_t)npencil + w - 1) / w;
          const size_t rounds = (blocks + per_cu * h->num_cu - 1) / (per_cu * h->num_cu);
          if (rounds < w_rounds) {
            w_rounds = rounds;
            W = w;
            n_glob = ng;
          }
        }
        if (w_rounds == 1) break;
      }
      if (W && w_rounds < best_rounds) {
        // (contiguous pencils: + the staging area of the wave-cooperative loads and stores)
        const size_t lds = (size_t)W * (parked - n_glob) * sizeof(T) +
                           (axis == 2 ? TileIO<T, U>::stage_elems * sizeof(T) : 0);
        const unsigned blocks = ((npencil + W - 1) / W + 7) / 8 * 8;
        uint32_t n_inner;
        size_t outer_stride, inner_stride, stride;
        if (axis == 2) { n_inner = npencil; outer_stride = 0; inner_stride = n; stride = 1; }
        else if (axis == 1) { n_inner = m[2]; outer_stride = (size_t)m[1] * m[2]; inner_stride = 1; stride = m[2]; }
        else { n_inner = m[1] * m[2]; outer_stride = batch_stride; inner_stride = 1; stride = (size_t)m[1] * m[2]; }
#define MGH_STREAM(CONTIG, PD)                                                                \
  {                                                                                           \
    static std::atomic<uint64_t> once{0};                                                     \
    TRY(allow_big_lds_once(k_ipk_stream<T, U, KR, PD, CONTIG, false>, once));                 \
    return launch(h, name, s, [&] {                                                           \
      k_ipk_stream<T, U, KR, PD, CONTIG, false><<<blocks, 64, lds, s>>>(                      \
          npencil, n_inner, outer_stride, inner_stride, stride, n, W, n_glob, x, tt, add_to,  \
          sign);                                                                              \
    });                                                                                       \
  }
        const int pd = h->ipk_pd;
        if (axis == 2) MGH_STREAM(true, 1)
        if (pd == 4) MGH_STREAM(false, 4)
        if (pd == 2) MGH_STREAM(false, 2)
        MGH_STREAM(false, 1)
#undef MGH_STREAM
      }
    }
    return kNotApplicable;
  };
  if constexpr (sizeof(T) == 4) {
    if (h->ipk_kr16 && n / U >= 32) {
      const int rc = stream_plan(std::integral_constant<uint32_t, 16>{}, 4);
      if (rc != kNotApplicable) return rc;
    }
  }
  {
    const int rc = stream_plan(std::integral_constant<uint32_t, 8>{}, 16);
    if (rc != kNotApplicable) return rc;
  }
  if (axis == 2 && best_w) {
    const uint32_t pad = (n % 2 == 0) ? 1u : 0u;
    const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + n - 1) / n);  // e/n for e < 2^17
    static std::atomic<uint64_t> once{0};
    TRY(allow_big_lds_once(k_ipk_lds_contig<T>, once));
    const uint32_t P = (uint32_t)best_w;
    return launch(h, name, s, [&] {
      k_ipk_lds_contig<T><<<(npencil + P - 1) / P, 256, P * pencil_bytes, s>>>(
          npencil, n, pad, magic, P, x, tt, add_to, sign, 0u);
    });
  }
  if (axis != 2 && best_w) {
    const uint32_t n_outer = axis == 1 ? m[0] : nbatch;
    const uint32_t n_inner = axis == 1 ? m[2] : m[1] * m[2];
    const size_t outer_stride = axis == 1 ? (size_t)m[1] * m[2] : batch_stride;
    const size_t stride = axis == 1 ? (size_t)m[2] : (size_t)m[1] * m[2];
    const size_t lds = best_w * pencil_bytes;
    const unsigned blocks = ((npencil + best_w - 1) / best_w + 7) / 8 * 8;  // XCD-contiguous tile ranges
#define MGH_STRIDED(W)                                                                        \
  {                                                                                           \
    static std::atomic<uint64_t> once{0};                                                                 \
    TRY(allow_big_lds_once(k_ipk_lds_strided<T, W>, once));                  \
    return launch(h, name, s, [&] {                                                           \
      k_ipk_lds_strided<T, W><<<blocks, 256, lds, s>>>(n_outer, n_inner, outer_stride,        \
                                                       stride, n, x, tt, add_to, sign);       \
    });                                                                                       \
  }
    if (best_w == 64) MGH_STRIDED(64)
    if (best_w == 48) MGH_STRIDED(48)
    if (best_w == 32) MGH_STRIDED(32)
    MGH_STRIDED(16)
#undef MGH_STRIDED
  }
  // pencils too long for LDS. Not too many of them (a 4194304 x 9 array: 9 strided pencils of
  // 2^21 elements per level; 100 x 100 x 6000: 2601 contiguous ones of 3001): in verified chunks
  if (spec_ok && n >= 2048 && npencil <= h->ipk_spec_max) return spec_solve();
  // ... else one thread per pencil straight from global memory
  if (nbatch > 1) {
    for (uint32_t bi = 0; bi < nbatch; bi++)
      TRY(ipk_launch<T>(h, axis, m, x + (size_t)bi * batch_stride, tt,
                        add_to ? add_to + (size_t)bi * batch_stride : nullptr, sign, s));
    return MGH_SUCCESS;
  }
  const dim3 pb(64, 1, 1);
  if (axis == 2) {
    const dim3 g((m[1] + 63) / 64, m[0], 1);
    return launch(h, name, s, [&] {
      k_ipk<T, 2><<<g, pb, 0, s>>>(m[0], m[1], m[2], x, tt, add_to, sign);
    });
  } else if (axis == 1) {
    const dim3 g((m[2] + 63) / 64, m[0], 1);
    return launch(h, name, s, [&] {
      k_ipk<T, 1><<<g, pb, 0, s>>>(m[0], m[1], m[2], x, tt, add_to, sign);
    });
  }
  const dim3 g((m[2] + 63) / 64, m[1], 1);
  return launch(h, name, s, [&] {
    k_ipk<T, 0><<<g, pb, 0, s>>>(m[0], m[1], m[2], x, tt, add_to, sign);
  });
}

// f-solve and c-solve of a compact (m0, m1, m2) box: one launch when a coarse plane fits in LDS
// (every level but the biggest), else the two tiled kernels.
template <typename T>
int ipk_fc_launch(mgh_hierarchy *h, const uint32_t *m, T *x, const T *tt_f, const T *tt_c,
                  hipStream_t s) {
  const uint32_t pitch = m[2] | 1u;
  if ((size_t)m[1] * pitch * sizeof(T) <= 150 * 1024 && m[1] <= 1024 && m[2] <= 1024) {
    static std::atomic<uint64_t> once{0};
    TRY(allow_big_lds_once(k_ipk_plane_fc<T>, once));
    const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + m[2] - 1) / m[2]);  // e / m2, e < 2^32 / m2
    return launch(h, "ipk_fc", s, [&] {
      k_ipk_plane_fc<T><<<m[0], 256, (size_t)m[1] * pitch * sizeof(T), s>>>(m[1], m[2], pitch, magic,
                                                                              x, tt_f, tt_c);
    });
  }
  TRY(ipk_launch<T>(h, 2, m, x, tt_f, nullptr, +1, s));
  return ipk_launch<T>(h, 1, m, x, tt_c, nullptr, +1, s);
}

// ---- correction = IPK(LPK(coefficients)) then +/- into nodal[l-1] --------------
// CalcCorrection3D (Correction/CalcCorrection3D.hpp:26-185) + AddND/SubtractND.
template <typename T>
int correction(mgh_hierarchy *h, int l, const T *coef, size_t cI, size_t cJ, T *target, int sign,
               hipStream_t s) {
  auto *ds = DS<T>(h);
  const LevelTables<T> &t = ds->lt[l];
  const Box3 &b = t.box;
  const dim3 blk(64, 4, 1);
  if (!ds->t1) {
    const Box3 &tb = ds->lt[h->L].box;
    TRY(dev_alloc(h, &ds->t1, (size_t)tb.n[0] * tb.n[1] * tb.m[2]));
    TRY(dev_alloc(h, &ds->t2, (size_t)tb.n[0] * tb.m[1] * tb.m[2]));
  }
  // LPK1 along f: (nr, nc, nf) -> (nr, nc, ff)
  TRY(launch(h, "lpk_f", s, [&] {
    k_lpk<T, 2><<<grid3(b.n[0], b.n[1], b.m[2], blk), blk, 0, s>>>(
        b.n[0], b.n[1], b.n[2], b.n[2], b.m[2], coef, cI, cJ, ds->t1, (size_t)b.n[1] * b.m[2],
        (size_t)b.m[2], t.mass[2], b.m[0], b.m[1]);
  }));
  T *cur = ds->t1;
  if (t.active[1]) {
    TRY(launch(h, "lpk_c", s, [&] {
      k_lpk<T, 1><<<grid3(b.n[0], b.m[1], b.m[2], blk), blk, 0, s>>>(
          b.n[0], b.n[1], b.m[2], b.n[1], b.m[1], cur, (size_t)b.n[1] * b.m[2], (size_t)b.m[2],
          ds->t2, (size_t)b.m[1] * b.m[2], (size_t)b.m[2], t.mass[1], 0, 0);
    }));
    cur = ds->t2;
  }
  if (t.active[0]) {
    TRY(launch(h, "lpk_r", s, [&] {
      k_lpk<T, 0><<<grid3(b.m[0], b.m[1], b.m[2], blk), blk, 0, s>>>(
          b.n[0], b.m[1], b.m[2], b.n[0], b.m[0], cur, (size_t)b.m[1] * b.m[2], (size_t)b.m[2],
          ds->t3, (size_t)b.m[1] * b.m[2], (size_t)b.m[2], t.mass[0], 0, 0);
    }));
    cur = ds->t3;
  }
  // IPK along f, c, r; the last one applies the correction to `target`
  const int last = t.active[0] ? 0 : (t.active[1] ? 1 : 2);
  TRY(ipk_launch<T>(h, 2, b.m, cur, t.thomas[2], last == 2 ? target : nullptr, sign, s));
  if (t.active[1])
    TRY(ipk_launch<T>(h, 1, b.m, cur, t.thomas[1], last == 1 ? target : nullptr, sign, s));
  if (t.active[0]) TRY(ipk_launch<T>(h, 0, b.m, cur, t.thomas[0], target, sign, s));
  return MGH_SUCCESS;
}

// Quantizer constants on the host (LinearQuantization.hpp:495-545 + volumes :186-195)
template <typename T> struct QuantParams {
  std::vector<T> qz, vol;  // per level
  int64_t dict_size = 0;
  int prep_huffman = 0;
  int64_t *q = nullptr;
  uint16_t *q16 = nullptr;  // instead of q: 16-bit dictionary symbols (mgh_decompose_quantize_sym16)
  unsigned long long *ocount = nullptr;
  uint64_t *oidx = nullptr;
  int64_t *oval = nullptr;
  unsigned long long ocap = 0;
  const T *d_qp = nullptr;  // device table [2 * (L + 1)] (k_make_qparams) instead of qz / vol
  // MGH_INLINE_QP: the call's quantizer constants in device memory + the reduction scalar; a finest
  // level that runs k_level_fused2 computes its quantizer itself and `after_first` is not called
  const QParamArgs<T> *d_qinl = nullptr;
  const unsigned long long *d_qslot = nullptr;
  std::function<int()> inline_done;  // host bookkeeping of the skipped launch
};

template <typename T>
QuantParams<T> make_quant_params(mgh_hierarchy *h, int ebtype, double tol, double s, double norm,
                                 bool reciprocal) {
  auto *hh = HH<T>(h);
  QuantParams<T> qp;
  qp.qz.resize(h->L + 1);
  qp.vol.resize(h->L + 1);
  hh->quantizers(ebtype, (T)tol, (T)s, (T)norm, reciprocal, qp.qz.data());
  const bool calc_vol = !((T)s == std::numeric_limits<T>::infinity());
  for (int l = 0; l <= h->L; l++) qp.vol[l] = calc_vol ? hh->level_volume(l, !reciprocal) : (T)1;
  return qp;
}

// Size class of a level for the fused kernels: 2 = plenty of tiles (long marches, RCH = 16),
// 1 = mid-size (RCH = 4), 0 = few tiles (one coarse plane per workgroup).
// nz: t-slices a launch of the D = 4 path covers (their workgroups count like tiles: 8 x 16395 x 39 x 39
// f64 has 3 tiles a slice and ran its 16395 planes in marches of 4 -- 5.6 ms, 4.6 with marches of 16).
inline int level_class(const mgh_hierarchy *h, const Box3 &b, size_t nz = 1) {
  constexpr int TC = 8, TF = 32;
  const size_t gx = (b.m[2] + TF - 1) / TF, gy = (b.m[1] + TC - 1) / TC;
  if (gx * gy * nz * ((b.m[0] + 15) / 16) >= h->cls2) return 2;
  if (gx * gy * nz * ((b.m[0] + 3) / 4) >= h->cls1) return 1;
  return 0;
}

// Coarse planes per workgroup of the fused level kernel (the kernel is compiled for up to 16).
inline int fused_rch(const mgh_hierarchy *h, int cls) { return h->rch[cls]; }

// r-chunks of a level on the fused kernel: chunks of rch coarse planes, the last one takes what is
// left (one plane more for sizes 2^k + 1)
constexpr unsigned kXcdRangeMinTiles = 64;  // (below: at most 8 tiles an XCD, one more or less is 12 % and more)
inline int fused_nchunk(int m_r, int rch) { return std::max(1, (m_r - 1 + rch - 1) / rch); }

// Level loop on the fused kernels (3 active dims): per level one fused
// coefficient/quantize/load-vector pass, three Thomas solves (the last one adds
// the correction into the coarse nodal array), then the head.
// `after_first` (the quantizer set-up) is issued right in front of the first launch that needs it.
// One level on the second-generation fused kernel (kernels_fused2.hpp). Tiles of the launch
// (Fused2Grid): a remainder of up to 4 coarse columns / rows beyond the full tiles goes to face
// tiles, the last r-chunk owns what is left of the planes (one more than the others for sizes
// 2^k + 1).
template <typename T, int OUTK, int TC, int TF, bool AGG = false>
int launch_fused2_t(mgh_hierarchy *h, const FusedArgs<T> &A, const Box3 &b, int cls, const char *nm,
                    hipStream_t s) {
  const int RCHv = fused_rch(h, cls);
  Fused2Grid G{};
  G.rch = RCHv;
  const int mfi = (int)b.m[2], mci = (int)b.m[1], mri = (int)b.m[0];
  const int nfull_f = (mfi - 1) / TF, rem_f = mfi - nfull_f * TF;
  const int nfull_c = (mci - 1) / TC, rem_c = mci - nfull_c * TC;
  const bool face_f = h->fused_faces && nfull_f >= 1 && rem_f <= 4;
  const bool face_c = h->fused_faces && nfull_c >= 1 && rem_c <= 4;
  G.gxm = face_f ? nfull_f : (mfi + TF - 1) / TF;
  const int gym = face_c ? nfull_c : (mci + TC - 1) / TC;
  G.n_main = G.gxm * gym;
  G.ff_F0 = nfull_f * TF;
  G.n_ff = face_f ? (mci + 63) / 64 : 0;
  G.cf_C0 = nfull_c * TC;
  G.n_cf = face_c ? ((face_f ? G.ff_F0 : mfi) + 63) / 64 : 0;
  G.nchunk = fused_nchunk(mri, RCHv);
  G.chunk_hi = G.nchunk;
  const unsigned ntile = (unsigned)(G.n_main + G.n_ff + G.n_cf);
  // (contiguous tile ranges per XCD only where there are tiles to hand out -- a cross-section of three
  // tiles padded to eight put every workgroup that had work on XCDs 0..2 -- 16395 x 39 x 39 f64: top
  // level 778 us; without the ranges the r-chunks rotate the tiles over the XCDs)
  G.xcd_ranges = ntile >= (h->fused_xcd == 2 ? 32u : kXcdRangeMinTiles) ? h->fused_xcd : 0;
  const dim3 grid(G.xcd_ranges ? (ntile + 7) / 8 * 8 : ntile, (unsigned)G.nchunk, 1);
  const bool faces = G.n_ff || G.n_cf;
#define MGH_F2(RCH)                                                                           \
  if (faces)                                                                                  \
    return launch_kernel(h, nm, s, k_level_fused2<T, OUTK, TC, TF, RCH, true, 0, AGG>, grid, dim3(256), 0, A, G, \
                         Fused4<T>{});                                                        \
  return launch_kernel(h, nm, s, k_level_fused2<T, OUTK, TC, TF, RCH, false, 0, AGG>, grid, dim3(256), 0, A, G, \
                       Fused4<T>{});
  MGH_F2(16)
#undef MGH_F2
}

// Tile shape: long marches (class 2) run 4 x 64 coarse nodes per tile -- every row a wave reads or
// writes is 512 contiguous bytes instead of 256, which the memory system rewards more than the
// larger halo (1.41 x instead of 1.24 x re-read) costs: top level of 512^3 f32 435 -> 383 us, same
// box, alternating runs. The short marches of the lower levels are a few us faster on 8 x 32.
// (... where the rows are long enough to fill them: 33 coarse nodes along f are one 8 x 32 tile and a
// face tile, or half a 4 x 64 tile)
inline bool fused_wide_tiles(const mgh_hierarchy *h, int cls, uint32_t mf) {
  if (h->fused_wide >= 2) return true;
  if (h->fused_wide != 1 || cls != 2) return false;
  auto filled = [&](uint32_t tf) {  // share of a main tile's columns that hold nodes
    const uint32_t nfull = (mf - 1) / tf, rem = mf - nfull * tf;
    if (h->fused_faces && nfull >= 1 && rem <= 4) return 1.0;  // (the remainder goes to a face tile)
    return (double)mf / (double)((mf + tf - 1) / tf * tf);
  };
  return filled(64) + 0.1 >= filled(32);
}
// A short FASTEST extent (AoS-like data: 2048 x 2048 x 17, 512^3 x 5): nine coarse nodes along f fill a
// quarter of an 8 x 32 tile's lanes, three of them a tenth. Tiles of 64 x 4 coarse nodes there -- the face
// tiles' shape as the main one (the (c, f) plane of such a level is nearly contiguous in memory, the short
// rows cost little). MGH_FUSED_TALL=0: never.
inline bool fused_tall_tiles(const mgh_hierarchy *h, const Box3 &b) {
  return h->fused_tall && b.m[2] <= 16 && b.m[1] >= 48;
}
template <typename T, int OUTK, bool AGG = false>
int launch_fused2(mgh_hierarchy *h, const FusedArgs<T> &A, const Box3 &b, int cls, const char *nm,
                  hipStream_t s) {
  if (fused_tall_tiles(h, b)) return launch_fused2_t<T, OUTK, 64, 4, AGG>(h, A, b, cls, nm, s);
  if (fused_wide_tiles(h, cls, b.m[2]))
    return launch_fused2_t<T, OUTK, 4, 64, AGG>(h, A, b, cls, nm, s);
  return launch_fused2_t<T, OUTK, 8, 32, AGG>(h, A, b, cls, nm, s);
}

// Which variant of the level kernel: the one that asks for outlier slots per wave and plane
// (right when few values leave the dictionary) or per workgroup and pair step (OutlierShared in
// kernels_fused2.hpp: more registers and LDS traffic, but the requests no longer queue on the one
// counter when many do). The last kernel of every call leaves the call's outlier count in host
// memory; what is found there now -- from the previous call or an earlier one, no synchronisation
// -- decides.
template <typename T> bool outlier_agg_now(mgh_hierarchy *h, const QuantParams<T> *qp) {
  auto *ds = DS<T>(h);
  if (!qp || !qp->prep_huffman || !qp->ocount) return false;
  if (!ds->outliers_seen && h->outlier_agg == 2) {
    if (hipHostMalloc(&ds->outliers_seen, sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess)
      *ds->outliers_seen = 0;
    else
      ds->outliers_seen = nullptr, (void)hipGetLastError();
  }
  const unsigned long long seen =
      ds->outliers_seen ? *reinterpret_cast<volatile unsigned long long *>(ds->outliers_seen) : 0;
  // (more than 0.5 % of the values AND enough of them for the requests to queue: on a 65^3 block,
  // where most outliers come from the small levels' own kernels, the stash only costs -- 64 blocks
  // of 65^3: 15.3 vs 16.9 ms per mgh_compress)
  return h->outlier_agg == 1 ||
         (h->outlier_agg == 2 && seen * 200 > (unsigned long long)h->total && seen > 200000);
}

__global__ void k_publish_count(const unsigned long long *count, unsigned long long *seen) { *seen = *count; }

template <typename T, int OUT, typename AfterFirst>
int decompose_fused4(mgh_hierarchy *h, const T *data, T *coeff, const QuantParams<T> *qp,
                     hipStream_t s, AfterFirst &&after_first);

template <typename T, int OUT, typename AfterFirst>
int decompose_fused(mgh_hierarchy *h, const T *data, T *coeff, const QuantParams<T> *qp,
                    hipStream_t s, AfterFirst &&after_first) {
  if (h->D == 4) return decompose_fused4<T, OUT>(h, data, coeff, qp, s, after_first);
  auto *ds = DS<T>(h);
  const int L = h->L;
  const size_t fI = ds->full_I, fJ = ds->full_J;
  const T *src = data;
  size_t sI = ds->src_I ? ds->src_I : fI, sJ = ds->src_J ? ds->src_J : fJ;  // (pitched input: mgh_set_ld)

  FusedArgs<T> A{};
  A.coef = coeff;
  A.dI = fI;
  A.dJ = fJ;
  if (OUT == OUT_Q) {
    A.q = qp->q;
    A.q16 = qp->q16;
    A.dict_size = qp->dict_size;
    A.prep_huffman = qp->prep_huffman;
    A.outlier_count = qp->ocount;
    A.outlier_idx = qp->oidx;
    A.outlier_val = qp->oval;
    A.outlier_cap = qp->ocap;
    A.qp = qp->d_qp;
    A.nlev = L + 1;
  }
  const bool agg = OUT == OUT_Q && outlier_agg_now<T>(h, qp);
  // levels whose working set fits in one workgroup's LDS run inside the tail kernel
  constexpr size_t kTailLdsMax = 150 * 1024;
  int l_tail = 0;  // levels l_tail .. 1 go to the tail (0 = none)
  for (int l = std::min(L, kTailMaxLevels); l >= 1; l--) {
    // (the table block of the tail's levels, plus the level above whose solves it may run)
    const size_t tab = ds->lt_end[std::min(l + 1, L)];
    if ((tail_lds_elems(ds->lt[l].box) + tab) * sizeof(T) + tail_header_bytes<T>() <= kTailLdsMax) {
      l_tail = l;
      break;
    }
  }
  bool tail_pre = false;  // the tail kernel also runs the solves of level l_tail + 1
  for (int l = L; l > l_tail; l--) {
    const LevelTables<T> &t = ds->lt[l];
    const Box3 &b = t.box;
    for (int k = 0; k < 3; k++) {
      A.n[k] = (int)b.n[k];
      A.m[k] = (int)b.m[k];
      A.ratio[k] = t.ratio[k];
      A.mass[k] = t.mass[k];
    }
    A.u = src;
    A.uI = sI;
    A.uJ = sJ;
    A.coarse = ds->nodal[l - 1];
    A.load = ds->t3;
    A.level = l;
    if (OUT == OUT_Q && !qp->d_qp) {
      A.quantizer = qp->qz[l];
      A.volume = qp->vol[l];
    }
    const int cls = level_class(h, b);
    {
      const bool inl = OUT == OUT_Q && l == L && qp->d_qinl && !(cls < h->box);
      A.qinl = inl ? qp->d_qinl : nullptr;
      A.qslot = inl ? qp->d_qslot : nullptr;
      if (l == L) {
        if (inl) TRY(qp->inline_done());
        else TRY(after_first());
      }
      // (the level kernels test the dictionary range in 32 bits: the entry points send larger
      // dictionaries through decompose + quantize)
      if (OUT == OUT_Q && !(qp->dict_size >= 0 && qp->dict_size <= ((int64_t)1 << 30)))
        return fail(MGH_ERR_INVALID_ARGUMENT, "fused path: dict_size must be at most 2^30");
      if (cls < h->box) {
        // small level: no march (kernels_box.hpp)
        constexpr int BR = 4, BC = 4, BF = 8;
        const int bx = ((int)b.m[2] + BF - 1) / BF, by = ((int)b.m[1] + BC - 1) / BC,
                  bz = ((int)b.m[0] + BR - 1) / BR;
        TRY(launch(h, OUT == OUT_Q ? "level_box_q" : "level_box", s, [&] {
          k_level_box<T, OUT, BR, BC, BF><<<(unsigned)(bx * by * bz), 256, 0, s>>>(A, bx, by);
        }));
      } else {
        // long marches (RCH = 16: 9% r-halo) when there are plenty of tiles, short ones
        // (RCH = 4) on the small levels where the march length is pure latency
        const char *nm = cls == 2 ? (OUT == OUT_Q ? "level_fused_q" : "level_fused")
                                  : (OUT == OUT_Q ? "level_fused_q_small" : "level_fused_small");
        if (OUT == OUT_Q && agg) {  // (many outliers last time: slot requests per workgroup)
          if (A.prep_huffman && !A.q16 && h->fused_fixed)
            TRY((launch_fused2<T, OUT == OUT_Q ? OUT_QH : OUT, OUT == OUT_Q>(h, A, b, cls, nm, s)));
          else
            TRY((launch_fused2<T, OUT, OUT == OUT_Q>(h, A, b, cls, nm, s)));
        } else if (OUT == OUT_Q && A.prep_huffman && !A.q16 && h->fused_fixed)
          TRY((launch_fused2<T, OUT == OUT_Q ? OUT_QH : OUT>(h, A, b, cls, nm, s)));
        else
          TRY((launch_fused2<T, OUT>(h, A, b, cls, nm, s)));
      }
    }
    // (the level right above the tail leaves its three solves to the tail kernel, which needs the
    // box in LDS anyway)
    tail_pre = h->tail_solves && l == l_tail + 1 && l_tail >= 1;
    if (!tail_pre) {
      // A load vector that does not fit the 256 MB memory-side cache (1024^3: 540 MB) is solved
      // in ranges of r-planes, f then c per range (both are independent per plane): the c-solve
      // finds what the f-solve just wrote in the cache instead of in HBM. MGH_IPK_RANGE_MB = size of
      // a range (0 = never).
      const size_t box_b = (size_t)b.m[0] * b.m[1] * b.m[2] * sizeof(T);
      const size_t range_b = (size_t)h->ipk_range_mb << 20;
      if (range_b && box_b > 2 * range_b) {
        const uint32_t nrange = (uint32_t)((box_b + range_b - 1) / range_b);
        for (uint32_t k = 0; k < nrange; k++) {
          const uint32_t R_lo = (uint32_t)((uint64_t)b.m[0] * k / nrange), R_hi = (uint32_t)((uint64_t)b.m[0] * (k + 1) / nrange);
          const uint32_t ms[3] = {R_hi - R_lo, b.m[1], b.m[2]};
          TRY(ipk_fc_launch<T>(h, ms, ds->t3 + (size_t)R_lo * b.m[1] * b.m[2], t.thomas[2], t.thomas[1], s));
        }
      } else {
        TRY(ipk_fc_launch<T>(h, b.m, ds->t3, t.thomas[2], t.thomas[1], s));
      }
      TRY(ipk_launch<T>(h, 0, b.m, ds->t3, t.thomas[0], ds->nodal[l - 1], +1, s));
    }
    src = ds->nodal[l - 1];
    sJ = b.m[2];
    sI = (size_t)b.m[1] * b.m[2];
  }
  if (L <= l_tail) TRY(after_first());
  if (l_tail >= 1) {
    TailArgs<T> TA{};
    TA.nlevels = l_tail;
    TA.fine = src;
    TA.fI = sI;
    TA.fJ = sJ;
    if (tail_pre) {
      TA.pre_load = ds->t3;
      for (int k = 0; k < 3; k++) TA.pre_thomas[k] = ds->lt[l_tail + 1].thomas[k];
    }
    for (int l = l_tail; l >= 1; l--) {
      TailLevel<T> &tl = TA.lv[l_tail - l];
      const LevelTables<T> &t = ds->lt[l];
      tl.b = t.box;
      for (int k = 0; k < 3; k++) {
        tl.ratio[k] = t.ratio[k];
        tl.mass[k] = t.mass[k];
        tl.thomas[k] = t.thomas[k];
      }
      tl.level = l;
      if (OUT == OUT_Q && !qp->d_qp) {
        tl.quantizer = qp->qz[l];
        tl.volume = qp->vol[l];
      }
    }
    if (OUT == OUT_Q && !qp->d_qp) {
      TA.head_quantizer = qp->qz[0];
      TA.head_volume = qp->vol[0];
    }
    TA.out = A;
    TA.outliers_seen = OUT == OUT_Q && A.prep_huffman ? ds->outliers_seen : nullptr;
    const size_t tab = ds->lt_end[tail_pre ? l_tail + 1 : l_tail];
    TA.tab_base = ds->tables;
    TA.tab_count = (uint32_t)tab;
    const size_t lds = (tail_lds_elems(ds->lt[l_tail].box) + tab) * sizeof(T) + tail_header_bytes<T>();
    static std::atomic<uint64_t> once{0};
    TRY(allow_big_lds_once(k_tail<T, OUT>, once));
    TRY(launch(h, "tail", s, [&] { k_tail<T, OUT><<<1, 1024, lds, s>>>(TA); }));
  } else {
    const Box3 &b = ds->lt[1].box;
    if (OUT == OUT_Q && !qp->d_qp) {
      A.quantizer = qp->qz[0];
      A.volume = qp->vol[0];
    }
    TRY(launch(h, "head_out", s, [&] {
      // (a hierarchy with one long and two short dimensions has few levels and a long head)
      const size_t tot = (size_t)b.m[0] * b.m[1] * b.m[2];
      k_head_out<T, OUT><<<(unsigned)std::min<size_t>((tot + 255) / 256, 1024), 256, 0, s>>>(
          (int)b.m[0], (int)b.m[1], (int)b.m[2], ds->nodal[0], A);
    }));
    if (OUT == OUT_Q && A.prep_huffman && A.outlier_count && ds->outliers_seen)
      k_publish_count<<<1, 1, 0, s>>>(A.outlier_count, ds->outliers_seen);
  }
  return MGH_SUCCESS;
}

// D = 4 on the 3-D tile code (kernels_fused2.hpp: TMODE 1 / 2, k_tsweep): per level the even
// slices of the slowest dimension t run the 3-D pass of the slice, the odd slices the TODD
// variant that interpolates across t as well; the fourth mass/restriction sweep and the four
// Thomas solves (f, c, r, t; the last one adds the correction into the coarse array) follow on
// the N/16-sized arrays. Order of every operation as in CalcCoefficientsND.hpp:25-236 and
// CalcCorrectionND.hpp:25-267 (dims D-1 .. 0): bit-identical to the generic N-D kernels.
// D = 4: the even and the odd slices of one level (kernels_fused2.hpp, TMODE 1 / 2)
template <typename T, int OUT, int TC, int TF, bool AGG = false>
int launch_fused4_t(mgh_hierarchy *h, const FusedArgs<T> &A, const Fused4<T> &Q, const Box3 &b, int cls,
                    int n_t, int m_t, hipStream_t s) {
    const int RCHv = fused_rch(h, cls);
    Fused2Grid G{};
    G.rch = RCHv;
    const int mfi = (int)b.m[2], mci = (int)b.m[1], mri = (int)b.m[0];
    const int nfull_f = (mfi - 1) / TF, rem_f = mfi - nfull_f * TF;
    const int nfull_c = (mci - 1) / TC, rem_c = mci - nfull_c * TC;
    const bool face_f = h->fused_faces && nfull_f >= 1 && rem_f <= 4;
    const bool face_c = h->fused_faces && nfull_c >= 1 && rem_c <= 4;
    G.gxm = face_f ? nfull_f : (mfi + TF - 1) / TF;
    const int gym = face_c ? nfull_c : (mci + TC - 1) / TC;
    G.n_main = G.gxm * gym;
    G.ff_F0 = nfull_f * TF;
    G.n_ff = face_f ? (mci + 63) / 64 : 0;
    G.cf_C0 = nfull_c * TC;
    G.n_cf = face_c ? ((face_f ? G.ff_F0 : mfi) + 63) / 64 : 0;
    G.nchunk = fused_nchunk(mri, RCHv);
    G.chunk_hi = G.nchunk;
    const unsigned ntile = (unsigned)(G.n_main + G.n_ff + G.n_cf);
    G.xcd_ranges = ntile >= (h->fused_xcd == 2 ? 32u : kXcdRangeMinTiles) ? h->fused_xcd : 0;  // (see launch_fused2_t)
    const unsigned gx = G.xcd_ranges ? (ntile + 7) / 8 * 8 : ntile;
    const bool faces = G.n_ff || G.n_cf;
    const unsigned n_even = (unsigned)m_t, n_odd = (unsigned)(n_t - m_t);
#define MGH_F4(RCH, TMODE, NZ, NAME)                                                          \
  if ((NZ) > 0) {                                                                             \
    const dim3 grid(gx, (unsigned)G.nchunk, (NZ));                                            \
    if (faces)                                                                                \
      TRY(launch_kernel(h, NAME, s, k_level_fused2<T, OUT, TC, TF, RCH, true, TMODE, AGG>, grid, dim3(256), 0, A, G, Q));  \
    else                                                                                      \
      TRY(launch_kernel(h, NAME, s, k_level_fused2<T, OUT, TC, TF, RCH, false, TMODE, AGG>, grid, dim3(256), 0, A, G, Q)); \
  }
    MGH_F4(16, 1, n_even, "level4_even")
    MGH_F4(16, 2, n_odd, "level4_odd")
#undef MGH_F4
    return MGH_SUCCESS;
}

// D = 4: the mass/restriction sweep along t of the per-slice load vectors: every input slice read
// once where the pencils are short enough for registers (k_tsweep_once), else slice by slice.
template <typename T>
int tsweep_launch(mgh_hierarchy *h, const T *load, T *corr, size_t M, int m_t, const T *mass, hipStream_t s) {
  constexpr int MT = 9;
  if (m_t <= MT) {
    const unsigned blocks = (unsigned)std::min<size_t>((M + 255) / 256, (size_t)h->num_cu * 32);
    return launch(h, "tsweep", s, [&] { k_tsweep_once<T, MT><<<blocks, 256, 0, s>>>(load, corr, M, m_t, mass); });
  }
  const dim3 grid((unsigned)std::min<size_t>((M + 255) / 256, 4096), (unsigned)m_t, 1);
  return launch(h, "tsweep", s, [&] { k_tsweep<T><<<grid, 256, 0, s>>>(load, corr, M, m_t, mass); });
}

// D = 4: Thomas solve along t of the correction (m_t, M) with the result added to / subtracted
// from the coarse array: short pencils go through registers (k_tsolve_apply), others through the
// generic strided solve.
template <typename T>
int tsolve_apply(mgh_hierarchy *h, T *corr, T *coarse, size_t M, int m_t, size_t m_rc, size_t m_f,
                 const T *tt, int sign, hipStream_t s) {
  constexpr int MT = 9;
  if (m_t <= MT) {
    const unsigned blocks = (unsigned)std::min<size_t>((M + 255) / 256, (size_t)h->num_cu * 32);
    return launch(h, "ipk_t", s, [&] {
      k_tsolve_apply<T, MT><<<blocks, 256, 0, s>>>(corr, coarse, M, m_t, tt, sign);
    });
  }
  const uint32_t m3t[3] = {(uint32_t)m_t, (uint32_t)m_rc, (uint32_t)m_f};
  return ipk_launch<T>(h, 0, m3t, corr, tt, coarse, sign, s);
}

// D = 4 work arrays: compact nodal arrays of the levels below the top, per-slice load vectors
// (padded positions of t) and the correction of the biggest coarse box
template <typename T> int ensure_state4(mgh_hierarchy *h) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const int L = h->L;
  const auto &sh = hh->level_shape;
  if (ds->state4_ready) return MGH_SUCCESS;
  // a failed attempt (out of memory on a big slab) leaves nothing behind -- neither memory nor its
  // share of mgh_device_bytes(): the next call starts over
  const size_t M = (size_t)sh[L - 1][1] * sh[L - 1][2] * sh[L - 1][3];
  auto nodal_count = [&](int l) { return (size_t)sh[l][0] * sh[l][1] * sh[l][2] * sh[l][3]; };
  const size_t load_count = (2 * (size_t)sh[L - 1][0] - 1) * M, corr_count = (size_t)sh[L - 1][0] * M;
  auto drop = [&] {
    for (size_t l = 0; l < ds->nodal4.size(); l++) dev_free(h, &ds->nodal4[l], nodal_count((int)l));
    ds->nodal4.clear();
    dev_free(h, &ds->load4, load_count);
    dev_free(h, &ds->corr4, corr_count);
  };
  drop();
  ds->nodal4.assign(L + 1, nullptr);
  int rc = MGH_SUCCESS;
  for (int l = 0; l < L && rc == MGH_SUCCESS; l++) rc = dev_alloc(h, &ds->nodal4[l], nodal_count(l));
  if (rc == MGH_SUCCESS) rc = dev_alloc(h, &ds->load4, load_count);
  if (rc == MGH_SUCCESS) rc = dev_alloc(h, &ds->corr4, corr_count);
  if (rc != MGH_SUCCESS) {
    drop();
    return rc;
  }
  ds->state4_ready = true;
  return MGH_SUCCESS;
}

template <typename T, int OUT, typename AfterFirst>
int decompose_fused4(mgh_hierarchy *h, const T *data, T *coeff, const QuantParams<T> *qp,
                     hipStream_t s, AfterFirst &&after_first) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const int L = h->L;
  const auto &sh = hh->level_shape;  // [l][d]
  const size_t full[4] = {(size_t)sh[L][1] * sh[L][2] * sh[L][3], (size_t)sh[L][2] * sh[L][3],
                          (size_t)sh[L][3], 1};
  TRY(ensure_state4<T>(h));
  FusedArgs<T> A{};
  A.coef = coeff;
  A.dI = full[1];
  A.dJ = full[2];
  if (OUT == OUT_Q) {
    A.q = qp->q;
    A.q16 = qp->q16;
    A.dict_size = qp->dict_size;
    A.prep_huffman = qp->prep_huffman;
    A.outlier_count = qp->ocount;
    A.outlier_idx = qp->oidx;
    A.outlier_val = qp->oval;
    A.outlier_cap = qp->ocap;
    A.qp = qp->d_qp;
    A.nlev = L + 1;
  }
  TRY(after_first());
  const bool agg = OUT == OUT_Q && outlier_agg_now<T>(h, qp);
  const T *src = data;
  size_t sT = full[0], sI = full[1], sJ = full[2];
  for (int l = L; l >= 1; l--) {
    const auto &N = sh[l], &Mc = sh[l - 1];
    Box3 b;
    for (int k = 0; k < 3; k++) {
      b.n[k] = (uint32_t)N[1 + k];
      b.m[k] = (uint32_t)Mc[1 + k];
      A.n[k] = (int)N[1 + k];
      A.m[k] = (int)Mc[1 + k];
      A.ratio[k] = ds->nd[l].ratio[1 + k];
      A.mass[k] = ds->nd[l].mass[1 + k];
    }
    const size_t M = (size_t)Mc[1] * Mc[2] * Mc[3];
    const int n_t = (int)N[0], m_t = (int)Mc[0];
    A.u = src;
    A.uI = sI;
    A.uJ = sJ;
    A.coarse = ds->nodal4[l - 1];
    A.load = ds->load4;
    A.level = l;
    if (OUT == OUT_Q && !qp->d_qp) {
      A.quantizer = qp->qz[l];
      A.volume = qp->vol[l];
    }
    Fused4<T> Q{};
    Q.ratio_t = ds->nd[l].ratio[0];
    Q.uT = sT;
    Q.dT = full[0];
    Q.cT = M;
    Q.n_t = n_t;
    Q.m_t = m_t;
    // an even n_t has a ghost slice (padded position n_t - 1): its load vector is zero
    if (n_t % 2 == 0)
      HIP_TRY(hipMemsetAsync(ds->load4 + (size_t)(n_t - 1) * M, 0, M * sizeof(T), s));
    const int cls = level_class(h, b, (size_t)std::max(1, n_t - m_t));
    const bool wide = fused_wide_tiles(h, cls, b.m[2]), tall = fused_tall_tiles(h, b);
    if (OUT == OUT_Q && agg) {
      if (tall) TRY((launch_fused4_t<T, OUT, 64, 4, OUT == OUT_Q>(h, A, Q, b, cls, n_t, m_t, s)));
      else if (wide) TRY((launch_fused4_t<T, OUT, 4, 64, OUT == OUT_Q>(h, A, Q, b, cls, n_t, m_t, s)));
      else TRY((launch_fused4_t<T, OUT, 8, 32, OUT == OUT_Q>(h, A, Q, b, cls, n_t, m_t, s)));
    } else if (tall)
      TRY((launch_fused4_t<T, OUT, 64, 4>(h, A, Q, b, cls, n_t, m_t, s)));
    else if (wide)
      TRY((launch_fused4_t<T, OUT, 4, 64>(h, A, Q, b, cls, n_t, m_t, s)));
    else
      TRY((launch_fused4_t<T, OUT, 8, 32>(h, A, Q, b, cls, n_t, m_t, s)));
    // t-sweep, then the Thomas solves f, c, r, t on the coarse box (m_t, m_r, m_c, m_f)
    {
      TRY((tsweep_launch<T>(h, ds->load4, ds->corr4, M, m_t, ds->nd[l].mass[0], s)));
    }
    const uint32_t m3a[3] = {(uint32_t)(m_t * Mc[1]), (uint32_t)Mc[2], (uint32_t)Mc[3]};
    TRY(ipk_launch<T>(h, 2, m3a, ds->corr4, ds->nd[l].thomas[3], nullptr, +1, s));
    TRY(ipk_launch<T>(h, 1, m3a, ds->corr4, ds->nd[l].thomas[2], nullptr, +1, s));
    TRY(ipk_launch<T>(h, 0, b.m, ds->corr4, ds->nd[l].thomas[1], nullptr, +1, s, (uint32_t)m_t, M));
    TRY((tsolve_apply<T>(h, ds->corr4, ds->nodal4[l - 1], M, m_t, Mc[1] * Mc[2], Mc[3], ds->nd[l].thomas[0], +1, s)));
    src = ds->nodal4[l - 1];
    sT = M;
    sI = (size_t)Mc[2] * Mc[3];
    sJ = Mc[3];
  }
  // head: the level-0 nodal values
  {
    if (OUT == OUT_Q && !qp->d_qp) {
      A.quantizer = qp->qz[0];
      A.volume = qp->vol[0];
    }
    const auto &M0 = sh[0];
    const size_t tot = (size_t)M0[0] * M0[1] * M0[2] * M0[3];
    TRY(launch(h, "head_out", s, [&] {
      k_head_out4<T, OUT><<<(unsigned)std::min<size_t>((tot + 1023) / 1024, 1024), 1024, 0, s>>>(
          (int)M0[0], (int)M0[1], (int)M0[2], (int)M0[3], ds->nodal4[0], A, full[0]);
    }));
    if (OUT == OUT_Q && A.prep_huffman && A.outlier_count && ds->outliers_seen)
      k_publish_count<<<1, 1, 0, s>>>(A.outlier_count, ds->outliers_seen);
  }
  return MGH_SUCCESS;
}

template <typename T, int OUT>
int decompose_fused(mgh_hierarchy *h, const T *data, T *coeff, const QuantParams<T> *qp,
                    hipStream_t s) {
  return decompose_fused<T, OUT>(h, data, coeff, qp, s, [] { return (int)MGH_SUCCESS; });
}

// The fused kernels index inside an r-plane with 32-bit offsets (and the emit pass with 32-bit
// byte offsets): planes of 2^29 elements or more go through the one-thread-per-element kernels.
inline bool fused_ok(const mgh_hierarchy *h) {
  return h->D == 3 && h->L >= 1 && h->plane_elems < ((uint64_t)1 << 29);
}
// compression side: also D = 4 (decompose_fused4); arrays of < 2^32 elements per t-slice
inline bool fused4_ok(const mgh_hierarchy *h) {
  return h->D == 4 && h->fused4 && !h->force_nd && h->L >= 1 && h->plane_elems < ((uint64_t)1 << 29);
}
inline bool fusedc_ok(const mgh_hierarchy *h) { return fused_ok(h) || fused4_ok(h); }


// ---- N-D path (D = 4, 5): in place on `v` (full array, reordered as levels proceed) -------
template <typename T> int nd_ensure(mgh_hierarchy *h) {
  auto *ds = DS<T>(h);
  if (!ds->nd_w) {
    TRY(dev_alloc(h, &ds->nd_w, (size_t)h->total));
    TRY(dev_alloc(h, &ds->nd_a, (size_t)h->total));
    TRY(dev_alloc(h, &ds->nd_b, (size_t)h->total));
  }
  return MGH_SUCCESS;
}

template <typename T> NdBox nd_box(mgh_hierarchy *h, int l) {
  auto *hh = HH<T>(h);
  NdBox b{};
  b.D = h->D;
  uint64_t sacc = 1;
  for (int d = h->D - 1; d >= 0; d--) {
    b.n[d] = (uint32_t)hh->level_shape[l][d];
    b.m[d] = (uint32_t)hh->level_shape[l - 1][d];
    b.fs[d] = sacc;
    sacc *= hh->shape[d];
  }
  return b;
}

inline unsigned nd_grid(uint64_t total) {
  return (unsigned)std::min<uint64_t>((total + 255) / 256, 256 * 16);
}

// correction of level l from the reordered coefficients in v; returns the compact result
// the row-wise kernels' view of a level (dimensions right-aligned to kNd)
inline NdRowBox nd_row_box(const NdBox &b) {
  NdRowBox r{};
  const int sh = kNd - b.D;
  for (int k = 0; k < kNd; k++) {
    r.n[k] = r.m[k] = 1;
    r.fs[k] = r.ns[k] = 0;
  }
  uint64_t sacc = 1;
  for (int d = b.D - 1; d >= 0; d--) {
    r.n[d + sh] = b.n[d];
    r.m[d + sh] = b.m[d];
    r.fs[d + sh] = b.fs[d];
    r.ns[d + sh] = sacc;
    sacc *= b.n[d];
  }
  r.rows = 1;
  for (int k = 0; k < kNd - 1; k++) r.rows *= r.n[k];
  return r;
}
inline unsigned nd_row_grid(uint64_t rows) {
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((rows + 4 * kNdRowsPerWave - 1) / (4 * kNdRowsPerWave), 1u << 20));
}
// the coefficient kernel's view of a level; false: an offset does not fit 32 bits (k_nd_coeff then)
inline bool nd_coeff_box(const NdBox &b, uint64_t nn, NdCoeffBox *out) {
  const NdRowBox rb = nd_row_box(b);
  uint64_t span = 0;
  for (int k = 0; k < kNd; k++) span += (uint64_t)(rb.n[k] - 1) * rb.fs[k];
  // rows of a group: eight where that still leaves every SIMD a wave (8 x 8 x 64^3: 13.5 us on the
  // second level with eight, 16.9 us with two), fewer on the small levels (7.7 -> 4.5 us on the third)
  const uint64_t lines = (uint64_t)rb.n[0] * rb.n[1] * rb.n[2];
  uint32_t gsz = kNdRowsPerWave;
  while (gsz > 2 && lines * ((rb.n[3] + gsz - 1) / gsz) < 4 * 1024) gsz /= 2;
  const uint64_t gpl = (rb.n[3] + gsz - 1) / gsz;
  const uint64_t groups = lines * gpl;
  if (span >= ((uint64_t)1 << 31) || nn >= ((uint64_t)1 << 31) || groups >= ((uint64_t)1 << 31)) return false;
  NdCoeffBox c{};
  for (int k = 0; k < kNd; k++) {
    c.n[k] = rb.n[k];
    c.m[k] = rb.m[k];
    c.fs[k] = (uint32_t)rb.fs[k];
    c.ns[k] = (uint32_t)rb.ns[k];
  }
  c.gsz = gsz;
  c.gpl = (uint32_t)gpl;
  c.groups = (uint32_t)groups;
  *out = c;
  return true;
}
template <typename T>
int nd_coeff_launch(mgh_hierarchy *h, const NdBox &b, const NdTables<T> &tb, T *w, T *v, uint64_t nn, int mode,
                    hipStream_t st) {
  NdCoeffBox cb;
  if (h->nd_rows && b.fs[b.D - 1] == 1 && nd_coeff_box(b, nn, &cb)) {
    NdTables<T> ta{};
    for (int d = 0; d < b.D; d++) ta.ratio[d + kNd - b.D] = tb.ratio[d];
    const unsigned grid = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((cb.groups + 3) / 4, 1u << 20));
    return launch(h, "nd_coeff", st, [&] { k_nd_coeff_rows<T><<<grid, 256, 0, st>>>(cb, ta, w, v, mode); });
  }
  return launch(h, "nd_coeff", st, [&] { k_nd_coeff<T><<<nd_grid(nn), 256, 0, st>>>(b, tb, w, v, nn, mode); });
}

// the first sweep's view (along the fastest dimension, s.a = D - 1); false: too many rows for 32 bits
inline bool nd_fast_sweep(const NdSweep &s, NdFastSweep *out) {
  NdFastSweep f{};
  const int sh = kNd - s.D;
  for (int k = 0; k < kNd - 1; k++) {
    f.e[k] = 1;
    f.is[k] = 0;
    f.mc[k] = 1;
  }
  for (int d = 0; d < s.D - 1; d++) {
    f.e[d + sh] = s.e[d];
    f.is[d + sh] = s.is[d];
    f.mc[d + sh] = s.mc[d];
  }
  f.sf = s.is[s.D - 1];
  f.n = s.n;
  f.m = s.m;
  f.zero_all_coarse = s.zero_all_coarse;
  const uint64_t lines = (uint64_t)f.e[0] * f.e[1] * f.e[2];
  uint32_t gsz = kNdRowsPerWave;
  while (gsz > 2 && lines * ((f.e[3] + gsz - 1) / gsz) < 4 * 1024) gsz /= 2;  // (as nd_coeff_box)
  const uint64_t gpl = (f.e[3] + gsz - 1) / gsz;
  if (lines * gpl >= ((uint64_t)1 << 31)) return false;
  f.gsz = gsz;
  f.gpl = (uint32_t)gpl;
  f.groups = (uint32_t)(lines * gpl);
  *out = f;
  return true;
}

template <typename T>
int nd_correction(mgh_hierarchy *h, int l, const T *v, const NdBox &b, T **out, hipStream_t st) {
  auto *ds = DS<T>(h);
  const int D = h->D;
  NdSweep s{};
  s.D = D;
  for (int d = 0; d < D; d++) {
    s.e[d] = b.n[d];
    s.is[d] = b.fs[d];
    s.mc[d] = b.m[d];
  }
  const T *cur = v;
  T *bufs[2] = {ds->nd_a, ds->nd_b};
  int which = 0;
  for (int a = D - 1; a >= 0; a--) {
    s.a = a;
    s.n = b.n[a];
    s.m = b.m[a];
    s.zero_all_coarse = (a == D - 1) ? 1 : 0;
    uint64_t total = 1;
    for (int d = 0; d < D; d++) total *= (d == a ? s.m : s.e[d]);
    T *dst = bufs[which];
    NdFastSweep fs;
    uint64_t outer = 1, inner = 1;
    for (int d = 0; d < a; d++) outer *= s.e[d];
    for (int d = a + 1; d < D; d++) inner *= s.e[d];
    const uint64_t plane = (uint64_t)s.m * inner, tiles = (plane + 1023) / 1024;
    if (h->nd_rows && a < D - 1 && (uint64_t)s.n * inner < ((uint64_t)1 << 31) && outer * tiles < ((uint64_t)1 << 31)) {
      // (the compact result of the sweep before: a 3-D view is all there is to it)
      const NdMidSweep ms{(uint32_t)outer, s.n, s.m, (uint32_t)inner, (uint32_t)plane, (uint32_t)tiles};
      const unsigned grid = (unsigned)std::min<uint64_t>(outer * tiles, 1u << 20);
      TRY(launch(h, "nd_lpk", st, [&] { k_nd_lpk_mid<T><<<grid, 256, 0, st>>>(ms, cur, dst, ds->nd[l].mass[a]); }));
    } else if (h->nd_rows && a == D - 1 && nd_fast_sweep(s, &fs)) {
      const unsigned grid = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((fs.groups + 3) / 4, 1u << 20));
      TRY(launch(h, "nd_lpk", st, [&] { k_nd_lpk_fast<T><<<grid, 256, 0, st>>>(fs, cur, dst, ds->nd[l].mass[a]); }));
    } else if (h->nd_rows) {
      NdRowSweep rs{};
      const int sh = kNd - D;
      for (int k = 0; k < kNd; k++) {
        rs.eo[k] = 1;
        rs.is[k] = 0;
        rs.mc[k] = 1;
      }
      for (int d = 0; d < D; d++) {
        rs.eo[d + sh] = d == a ? s.m : s.e[d];
        rs.is[d + sh] = s.is[d];
        rs.mc[d + sh] = s.mc[d];
      }
      rs.a = a + sh;
      rs.n = s.n;
      rs.m = s.m;
      rs.zero_all_coarse = s.zero_all_coarse;
      rs.rows = 1;
      for (int k = 0; k < kNd - 1; k++) rs.rows *= rs.eo[k];
      TRY(launch(h, "nd_lpk", st, [&] {
        k_nd_lpk_rows<T><<<nd_row_grid(rs.rows), 256, 0, st>>>(rs, cur, dst, ds->nd[l].mass[a]);
      }));
    } else {
      TRY(launch(h, "nd_lpk", st, [&] {
        k_nd_lpk<T><<<nd_grid(total), 256, 0, st>>>(s, cur, dst, ds->nd[l].mass[a], total);
      }));
    }
    cur = dst;
    which ^= 1;
    s.e[a] = s.m;
    uint64_t sacc = 1;
    for (int d = D - 1; d >= 0; d--) {
      s.is[d] = sacc;
      sacc *= s.e[d];
    }
  }
  T *x = const_cast<T *>(cur);
  for (int a = D - 1; a >= 0; a--) {
    uint64_t np = 1;
    for (int d = 0; d < D; d++)
      if (d != a) np *= s.e[d];
    // the box is compact: a solve along dim a is the strided (or, a = D - 1, contiguous) solve of
    // the 3-D view (dims before a, a, dims behind a) -- the tuned kernels of ipk_launch (LDS-staged,
    // streaming, verified chunks for long pencils) instead of one thread per pencil walking global
    // memory (5 x 5 x 5 x 5 x 40000: 1.5 ms per launch). Same arithmetic, same bits.
    uint64_t outer = 1, inner = 1;
    for (int d = 0; d < a; d++) outer *= s.e[d];
    for (int d = a + 1; d < D; d++) inner *= s.e[d];
    const uint64_t na = s.e[a];
    if (!h->force_nd_ipk && outer * na * inner < ((uint64_t)1 << 31) && na >= 2) {
      if (a == D - 1) {
        const uint32_t m3[3] = {(uint32_t)outer, 1u, (uint32_t)na};
        TRY(ipk_launch<T>(h, 2, m3, x, ds->nd[l].thomas[a], nullptr, +1, st));
      } else {
        const uint32_t m3[3] = {(uint32_t)outer, (uint32_t)na, (uint32_t)inner};
        TRY(ipk_launch<T>(h, 1, m3, x, ds->nd[l].thomas[a], nullptr, +1, st));
      }
      continue;
    }
    TRY(launch(h, "nd_ipk", st, [&] {
      k_nd_ipk<T><<<nd_grid(np), 256, 0, st>>>(D, a, s, x, ds->nd[l].thomas[a], np);
    }));
  }
  *out = x;
  return MGH_SUCCESS;
}

// src_top: the input when it is another array than v (out of place): the top level, whose fine box
// IS the array, reads it where it is -- neither the copy into v nor the one into the natural-order
// work array happen (two passes over the data; every element of v is written by that level).
template <typename T> int decompose_nd(mgh_hierarchy *h, T *v, hipStream_t st, const T *src_top = nullptr) {
  auto *ds = DS<T>(h);
  TRY(nd_ensure<T>(h));
  if (src_top && !(h->L >= 1 && nd_box_is_whole_array(nd_box<T>(h, h->L)))) {
    HIP_TRY(hipMemcpyAsync(v, src_top, h->total * sizeof(T), hipMemcpyDeviceToDevice, st));
    src_top = nullptr;
  }
  for (int l = h->L; l >= 1; l--) {
    const NdBox b = nd_box<T>(h, l);
    NdTables<T> tb{};
    uint64_t nn = 1, mm = 1;
    for (int d = 0; d < h->D; d++) {
      tb.ratio[d] = ds->nd[l].ratio[d];
      nn *= b.n[d];
      mm *= b.m[d];
    }
    T *w = ds->nd_w;
    if (l == h->L && src_top) {
      w = const_cast<T *>(src_top);  // (mode 0 only reads it)
    } else if (nd_box_is_whole_array(b)) {  // the top level: the fine box IS the array -- a plain copy
      TRY(launch(h, "nd_gather", st, [&] {
        (void)hipMemcpyAsync(ds->nd_w, v, nn * sizeof(T), hipMemcpyDeviceToDevice, st);
      }));
    } else {
      TRY(launch(h, "nd_gather", st, [&] {
        k_nd_gather<T><<<nd_grid(nn), 256, 0, st>>>(b, v, ds->nd_w, nn, 0);
      }));
    }
    TRY(nd_coeff_launch<T>(h, b, tb, w, v, nn, 0, st));
    T *corr = nullptr;
    TRY(nd_correction<T>(h, l, v, b, &corr, st));
    TRY(launch(h, "nd_apply", st, [&] {
      k_nd_apply<T><<<nd_grid(mm), 256, 0, st>>>(b, corr, v, mm, +1);
    }));
  }
  return MGH_SUCCESS;
}

template <typename T> int recompose_nd(mgh_hierarchy *h, T *v, hipStream_t st) {
  auto *ds = DS<T>(h);
  TRY(nd_ensure<T>(h));
  for (int l = 1; l <= h->L; l++) {
    const NdBox b = nd_box<T>(h, l);
    NdTables<T> tb{};
    uint64_t nn = 1, mm = 1;
    for (int d = 0; d < h->D; d++) {
      tb.ratio[d] = ds->nd[l].ratio[d];
      nn *= b.n[d];
      mm *= b.m[d];
    }
    T *corr = nullptr;
    TRY(nd_correction<T>(h, l, v, b, &corr, st));
    TRY(launch(h, "nd_apply", st, [&] {
      k_nd_apply<T><<<nd_grid(mm), 256, 0, st>>>(b, corr, v, mm, -1);
    }));
    TRY(nd_coeff_launch<T>(h, b, tb, ds->nd_w, v, nn, 1, st));
    TRY(nd_coeff_launch<T>(h, b, tb, ds->nd_w, v, nn, 2, st));
    if (nd_box_is_whole_array(b)) {
      TRY(launch(h, "nd_gather", st, [&] {
        (void)hipMemcpyAsync(v, ds->nd_w, nn * sizeof(T), hipMemcpyDeviceToDevice, st);
      }));
    } else {
      TRY(launch(h, "nd_gather", st, [&] {
        k_nd_gather<T><<<nd_grid(nn), 256, 0, st>>>(b, v, ds->nd_w, nn, 1);
      }));
    }
  }
  return MGH_SUCCESS;
}

template <typename T>
int decompose_impl(mgh_hierarchy *h, const T *data, T *coeff, hipStream_t s) {
  auto *ds = DS<T>(h);
  if (fused4_ok(h) && !h->force_v1) {
    const T *src4 = data;
    if ((const void *)data == (const void *)coeff) {
      TRY(ensure_scratch<T>(h));
      HIP_TRY(hipMemcpyAsync(ds->scratch_full, data, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
      src4 = ds->scratch_full;
    }
    return decompose_fused<T, OUT_T>(h, src4, coeff, nullptr, s);
  }
  if (h->D > 3 || h->force_nd)
    return decompose_nd<T>(h, coeff, s, (const void *)data != (const void *)coeff ? data : nullptr);
  const int L = h->L;
  const size_t fI = ds->full_I, fJ = ds->full_J;
  const T *src = data;
  size_t sI = fI, sJ = fJ;
  if ((const void *)data == (const void *)coeff) {
    TRY(ensure_scratch<T>(h));
    HIP_TRY(hipMemcpyAsync(ds->scratch_full, data, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
    src = ds->scratch_full;
  }
  if (fused_ok(h) && !h->force_v1) return decompose_fused<T, OUT_T>(h, src, coeff, nullptr, s);
  const dim3 blk(64, 4, 1);
  for (int l = L; l >= 1; l--) {
    const LevelTables<T> &t = ds->lt[l];
    const Box3 &b = t.box;
    T *coarse = ds->nodal[l - 1];
    TRY(launch(h, "gpk_reo", s, [&] {
      k_gpk_reo<T><<<grid3(b.n[0], b.n[1], b.n[2], blk), blk, 0, s>>>(
          b, src, sI, sJ, coarse, coeff, fI, fJ, t.ratio[0], t.ratio[1], t.ratio[2]);
    }));
    TRY(correction<T>(h, l, coeff, fI, fJ, coarse, +1, s));
    src = coarse;
    sJ = b.m[2];
    sI = (size_t)b.m[1] * b.m[2];
  }
  // level-0 nodal values are the head of the coefficient array
  if (L >= 1) {
    const Box3 &b = ds->lt[1].box;
    TRY(launch(h, "copy_box", s, [&] {
      k_copy_box<T><<<grid3(b.m[0], b.m[1], b.m[2], blk), blk, 0, s>>>(
          b.m[0], b.m[1], b.m[2], ds->nodal[0], (size_t)b.m[1] * b.m[2], (size_t)b.m[2], coeff, fI,
          fJ);
    }));
  } else if ((const void *)data != (const void *)coeff) {
    HIP_TRY(hipMemcpyAsync(coeff, data, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
  }
  return MGH_SUCCESS;
}


template <typename T, typename QT, typename QTL = QT>
int recompose_levels(mgh_hierarchy *h, RecomposeArgs<T> A, const std::vector<T> &level_qv, T *data,
                     hipStream_t st, const RecomposeArgs<T> *AL = nullptr, int ntop = 1);

template <typename T, typename QT, typename QTL = QT>
int recompose_levels4(mgh_hierarchy *h, RecomposeArgs<T> A, const std::vector<T> &level_qv, T *data,
                      hipStream_t st, const RecomposeArgs<T> *AL = nullptr, size_t A_sT = 0, int ntop = 1);
inline bool fused4_ok(const mgh_hierarchy *h);

template <typename T>
int recompose_impl(mgh_hierarchy *h, const T *coeff, T *data, hipStream_t s) {
  auto *ds = DS<T>(h);
  if (fused4_ok(h) && !h->force_v1) {
    // D = 4 on the slice-by-slice level loop, reading floating-point coefficients
    const T *C = coeff;
    if ((const void *)data == (const void *)coeff) {
      TRY(ensure_scratch<T>(h));
      HIP_TRY(hipMemcpyAsync(ds->scratch_full, coeff, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
      C = ds->scratch_full;
    }
    RecomposeArgs<T> A{};
    A.coef = C;
    return recompose_levels4<T, T>(h, A, std::vector<T>(h->L + 1, (T)1), data, s);
  }
  if (h->D > 3 || h->force_nd) {
    if ((const void *)data != (const void *)coeff)
      HIP_TRY(hipMemcpyAsync(data, coeff, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
    return recompose_nd<T>(h, data, s);
  }
  const int L = h->L;
  const size_t fI = ds->full_I, fJ = ds->full_J;
  const T *C = coeff;
  if ((const void *)data == (const void *)coeff) {
    if (L == 0) return MGH_SUCCESS;
    TRY(ensure_scratch<T>(h));
    HIP_TRY(hipMemcpyAsync(ds->scratch_full, coeff, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
    C = ds->scratch_full;
  }
  if (L == 0) {
    HIP_TRY(hipMemcpyAsync(data, coeff, h->total * sizeof(T), hipMemcpyDeviceToDevice, s));
    return MGH_SUCCESS;
  }
  if (fused_ok(h) && !h->force_v1) {
    // the level loop of the fused decompression, reading floating-point coefficients
    RecomposeArgs<T> A{};
    A.coef = C;
    A.dI = fI;
    A.dJ = fJ;
    return recompose_levels<T, T>(h, A, std::vector<T>(L + 1, (T)1), data, s);
  }
  const dim3 blk(64, 4, 1);
  {
    const Box3 &b = ds->lt[1].box;
    TRY(launch(h, "copy_box", s, [&] {
      k_copy_box<T><<<grid3(b.m[0], b.m[1], b.m[2], blk), blk, 0, s>>>(
          b.m[0], b.m[1], b.m[2], C, fI, fJ, ds->nodal[0], (size_t)b.m[1] * b.m[2], (size_t)b.m[2]);
    }));
  }
  for (int l = 1; l <= L; l++) {
    const LevelTables<T> &t = ds->lt[l];
    const Box3 &b = t.box;
    T *coarse = ds->nodal[l - 1];
    TRY(correction<T>(h, l, C, fI, fJ, coarse, -1, s));
    T *out = (l == L) ? data : ds->nodal[l];
    const size_t oJ = (l == L) ? fJ : b.n[2];
    const size_t oI = (l == L) ? fI : (size_t)b.n[1] * b.n[2];
    TRY(launch(h, "gpk_rev", s, [&] {
      k_gpk_rev<T><<<grid3(b.n[0], b.n[1], b.n[2], blk), blk, 0, s>>>(
          b, coarse, C, fI, fJ, out, oI, oJ, t.ratio[0], t.ratio[1], t.ratio[2]);
    }));
  }
  return MGH_SUCCESS;
}


// Fused decompression: outlier restore, then per level (coarse to fine) the load vector
// straight from the quantized coefficients, three Thomas solves subtracting the correction
// from the coarse nodes, and the node restore with the dequantizer fused in
// (Compressor::Decompress lines 256-257 = Dequantize + Recompose). With QT = T the same level
// loop runs on floating-point coefficients (Compressor::Recompose on its own).
// Node restore of one level (or one t-slice of a 4-D level): the marching kernel
// (kernels_recompose2.hpp), or the row-pair / row kernels of kernels_recompose.hpp
// (MGH_RESTORE_V=2, MGH_RESTORE_ROWS=1: cross-checks).
// (TC x TF coarse nodes per workgroup: 4 x 64, or 64 x 4 where the fastest extent is short -- fused_tall_tiles)
template <typename T, typename QT, bool TODD, int TC, int TF>
int launch_restore3(mgh_hierarchy *h, const RecomposeArgs<T> &A, const Box3 &b, const char *nm, hipStream_t st) {
    Restore3Grid G{};
    G.gxm = ((int)b.m[2] + TF - 1) / TF;
    G.ntile = G.gxm * (((int)b.m[1] + TC - 1) / TC);
    // chunk length: long marches where there are plenty of tiles, short ones (more workgroups)
    // on the small levels -- a chunk costs one extra coarse plane of interpolants only
    const int nslice = A.zb_mode ? (A.zb_mode == 2 ? A.zb_mt : A.zb_nt - A.zb_mt) : 1;
    const int want = 2048;
    G.rch = std::max(1, std::min(16, (int)((int64_t)b.m[0] * G.ntile * nslice / want)));
    G.nchunk = ((int)b.m[0] + G.rch - 1) / G.rch;
    const dim3 grid((unsigned)G.ntile, (unsigned)G.nchunk, (unsigned)nslice);
    return launch(h, nm, st, [&] { k_level_restore3_q<T, QT, TODD, TC, TF><<<grid, TC * TF, 0, st>>>(A, G); });
}

template <typename T, typename QT, bool TODD>
int launch_restore(mgh_hierarchy *h, const RecomposeArgs<T> &A, const Box3 &b, const char *nm, hipStream_t st) {
  if (h->restore_v == 3 && !h->restore_rows && fused_tall_tiles(h, b))
    return launch_restore3<T, QT, TODD, 64, 4>(h, A, b, nm, st);
  if (h->restore_v == 3 && !h->restore_rows) return launch_restore3<T, QT, TODD, 4, 64>(h, A, b, nm, st);
  const dim3 blk(64, 4, 1);
  if (h->restore_rows && !TODD)
    return launch(h, nm, st, [&] {
      k_level_restore_q<T, QT><<<dim3(1, (b.n[1] + 3) / 4, b.n[0]), blk, 0, st>>>(A);
    });
  const dim3 grid(1, ((b.n[1] + 1) / 2 + 3) / 4, b.n[0]);
  return launch(h, nm, st, [&] { k_level_restore2_q<T, QT, TODD><<<grid, blk, 0, st>>>(A); });
}
// Load-vector pass of the decompression side (one level, or one t-slice of a 4-D level):
// one plane per step, march length by the number of tiles.
template <typename T, typename QT, int TC, int TF>
int launch_loadvec_t(mgh_hierarchy *h, const RecomposeArgs<T> &A, const Box3 &b, hipStream_t st) {
  const unsigned gx = (b.m[2] + TF - 1) / TF, gy = (b.m[1] + TC - 1) / TC;
  // (A.zb_mode == 1: all the padded t positions of a 4-D level in this launch, the r-chunks of one
  // behind those of the other in grid.z -- which holds 65535 at most: shorter marches only where
  // they fit)
  const unsigned ns = A.zb_mode == 1 ? (unsigned)(2 * A.zb_mt - 1) : 1u;
  const unsigned z16 = (b.m[0] + 15) / 16, z4 = (b.m[0] + 3) / 4, z1 = b.m[0];
  RecomposeArgs<T> B = A;
  if ((size_t)gx * gy * z16 * ns >= 2048 || (size_t)z4 * ns > 65535) {
    B.zb_nz = (int)z16;
    return launch(h, "loadvec_q", st, [&] {
      k_level_loadvec_q<T, QT, TC, TF, 16><<<dim3(gx, gy, z16 * ns), 256, 0, st>>>(B);
    });
  }
  if ((size_t)gx * gy * z4 * ns >= 256 || (size_t)z1 * ns > 65535) {
    B.zb_nz = (int)z4;
    return launch(h, "loadvec_q_small", st, [&] {
      k_level_loadvec_q<T, QT, TC, TF, 4><<<dim3(gx, gy, z4 * ns), 256, 0, st>>>(B);
    });
  }
  B.zb_nz = (int)z1;
  return launch(h, "loadvec_q_small", st, [&] {
    k_level_loadvec_q<T, QT, TC, TF, 1><<<dim3(gx, gy, z1 * ns), 256, 0, st>>>(B);
  });
}

// (8 x 32 coarse nodes per workgroup, or 64 x 4 where the fastest extent is short -- fused_tall_tiles)
template <typename T, typename QT>
int launch_loadvec(mgh_hierarchy *h, const RecomposeArgs<T> &A, const Box3 &b, hipStream_t st) {
  if (fused_tall_tiles(h, b)) return launch_loadvec_t<T, QT, 64, 4>(h, A, b, st);
  return launch_loadvec_t<T, QT, 8, 32>(h, A, b, st);
}

// QTL / AL: coefficient source of the `ntop` FINEST levels when it differs from that of the levels
// below (16-bit symbols for the finest levels, int64 of the coarse corner box for the rest:
// dequantize_recompose_fused16); AL == nullptr: one source for all levels.
template <typename T, typename QT, typename QTL>
int recompose_levels(mgh_hierarchy *h, RecomposeArgs<T> A, const std::vector<T> &level_qv, T *data,
                     hipStream_t st, const RecomposeArgs<T> *AL, int ntop) {
  auto *ds = DS<T>(h);
  const int L = h->L;
  // levels 1 .. l_head run inside ONE single-workgroup kernel (their working set fits in LDS);
  // MGH_NO_RECOMPOSE_HEAD=1: every level with its own launches (cross-check)
  const bool no_head = h->no_head;
  int l_head = 0;
  if (!no_head) {
    for (int l = 1; l <= std::min(AL ? L - ntop : L, kTailMaxLevels); l++) {
      if ((head_lds_elems(ds->lt[l].box) + ds->lt_end[l]) * sizeof(T) > 150 * 1024) break;
      l_head = l;
    }
  }
  if (l_head >= 1) {
    HeadArgs<T> HA{};
    HA.nlevels = l_head;
    for (int l = 1; l <= l_head; l++) {
      HeadLevel<T> &hl = HA.lv[l - 1];
      const LevelTables<T> &t = ds->lt[l];
      hl.b = t.box;
      for (int k = 0; k < 3; k++) {
        hl.ratio[k] = t.ratio[k];
        hl.mass[k] = t.mass[k];
        hl.thomas[k] = t.thomas[k];
      }
      hl.qv = level_qv[l];
    }
    HA.qv0 = level_qv[0];
    HA.in = A;
    const Box3 &bl = ds->lt[l_head].box;
    HA.out = (l_head == L) ? data : ds->nodal[l_head];
    HA.oJ = (l_head == L) ? (ds->dst_J ? ds->dst_J : ds->full_J) : bl.n[2];
    HA.oI = (l_head == L) ? (ds->dst_I ? ds->dst_I : ds->full_I) : (size_t)bl.n[1] * bl.n[2];
    HA.tab_base = ds->tables;
    HA.tab_count = (uint32_t)ds->lt_end[l_head];
    const size_t lds = (head_lds_elems(bl) + ds->lt_end[l_head]) * sizeof(T);
    static std::atomic<uint64_t> once{0};
    TRY(allow_big_lds_once(k_recompose_head<T, QT>, once));
    TRY(launch(h, "recompose_head", st, [&] { k_recompose_head<T, QT><<<1, 1024, lds, st>>>(HA); }));
  } else {
    const Box3 &b = ds->lt[1].box;
    A.qv = level_qv[0];
    TRY(launch(h, "head_in", st, [&] {
      const size_t tot = (size_t)b.m[0] * b.m[1] * b.m[2];
      k_head_in_q<T, QT><<<(unsigned)std::min<size_t>((tot + 255) / 256, 1024), 256, 0, st>>>(
          (int)b.m[0], (int)b.m[1], (int)b.m[2], A, ds->nodal[0]);
    }));
  }
  for (int l = l_head + 1; l <= L; l++) {
    const LevelTables<T> &t = ds->lt[l];
    const Box3 &b = t.box;
    const bool top = AL && l > L - ntop;
    RecomposeArgs<T> B = top ? *AL : A;
    for (int k = 0; k < 3; k++) {
      B.n[k] = (int)b.n[k];
      B.m[k] = (int)b.m[k];
      B.ratio[k] = t.ratio[k];
      B.mass[k] = t.mass[k];
    }
    B.qv = level_qv[l];
    B.load = ds->t3;
    B.coarse = ds->nodal[l - 1];
    if (top) TRY((launch_loadvec<T, QTL>(h, B, b, st)));
    else TRY((launch_loadvec<T, QT>(h, B, b, st)));
    TRY(ipk_fc_launch<T>(h, b.m, ds->t3, t.thomas[2], t.thomas[1], st));
    TRY(ipk_launch<T>(h, 0, b.m, ds->t3, t.thomas[0], ds->nodal[l - 1], -1, st));
    B.fine = (l == L) ? data : ds->nodal[l];
    B.fJ = (l == L) ? (ds->dst_J ? ds->dst_J : ds->full_J) : b.n[2];
    B.fI = (l == L) ? (ds->dst_I ? ds->dst_I : ds->full_I) : (size_t)b.n[1] * b.n[2];
    if (top) TRY((launch_restore<T, QTL, false>(h, B, b, "restore_q", st)));
    else TRY((launch_restore<T, QT, false>(h, B, b, "restore_q", st)));
  }
  return MGH_SUCCESS;
}

// D = 4, the mirror of decompose_fused4: per level (coarse to fine) the load vector of every
// t-slice with the 3-D kernel (odd slices: every node is a coefficient), the t-sweep, four Thomas
// solves subtracting the correction from the coarse nodes, then the node restore slice by slice
// (odd slices interpolate across t between the two neighbouring coarse slices).
// A_sT: element stride between two t-slices of A's source (0: the full array's). AL / QTL as in
// recompose_levels (the finest level's source has the full array's strides).
template <typename T, typename QT, typename QTL>
int recompose_levels4(mgh_hierarchy *h, RecomposeArgs<T> A0, const std::vector<T> &level_qv, T *data,
                      hipStream_t st, const RecomposeArgs<T> *AL, size_t A_sT, int ntop) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const int L = h->L;
  const auto &sh = hh->level_shape;
  const size_t full[4] = {(size_t)sh[L][1] * sh[L][2] * sh[L][3], (size_t)sh[L][2] * sh[L][3],
                          (size_t)sh[L][3], 1};
  TRY(ensure_state4<T>(h));
  if (!A_sT) {
    A_sT = full[0];
    A0.dI = full[1];
    A0.dJ = full[2];
  }
  {
    const auto &M0 = sh[0];
    const size_t tot = (size_t)M0[0] * M0[1] * M0[2] * M0[3];
    A0.qv = level_qv[0];
    TRY(launch(h, "head_in", st, [&] {
      k_head_in4_q<T, QT><<<(unsigned)std::min<size_t>((tot + 255) / 256, 1024), 256, 0, st>>>(
          (int)M0[0], (int)M0[1], (int)M0[2], (int)M0[3], A0, A_sT, ds->nodal4[0]);
    }));
  }
  for (int l = 1; l <= L; l++) {
    const bool top = AL && l > L - ntop;
    RecomposeArgs<T> A = top ? *AL : A0;
    const size_t sT = top ? full[0] : A_sT;
    if (top) {
      A.dI = full[1];
      A.dJ = full[2];
    }
    const auto &N = sh[l], &Mc = sh[l - 1];
    Box3 b;
    for (int k = 0; k < 3; k++) {
      b.n[k] = (uint32_t)N[1 + k];
      b.m[k] = (uint32_t)Mc[1 + k];
      A.n[k] = (int)N[1 + k];
      A.m[k] = (int)Mc[1 + k];
      A.ratio[k] = ds->nd[l].ratio[1 + k];
      A.mass[k] = ds->nd[l].mass[1 + k];
    }
    A.qv = level_qv[l];
    A.ratio_t = ds->nd[l].ratio[0];
    const size_t M = (size_t)Mc[1] * Mc[2] * Mc[3];
    const int n_t = (int)N[0], m_t = (int)Mc[0];
    // ---- load vectors of the padded t positions: one launch for all of them (grid.z limit
    // permitting; MGH_SLICE_BATCH=0: a launch per slice)
    const bool batch = h->slice_batch && h->restore_v == 3 && !h->restore_rows &&
                       (size_t)(2 * m_t - 1) * ((Mc[1] + 15) / 16) < 65536;  // (grid.z of the load-vector launch)
    if (n_t % 2 == 0)  // ghost slice
      HIP_TRY(hipMemsetAsync(ds->load4 + (size_t)(n_t - 1) * M, 0, M * sizeof(T), st));
    if (batch) {
      A.zb_mode = 1;
      A.zb_mt = m_t;
      A.zb_nt = n_t;
      A.zb_sT = sT;
      A.zb_M = M;
      A.load = ds->load4;
      if (top) TRY((launch_loadvec<T, QTL>(h, A, b, st)));
      else TRY((launch_loadvec<T, QT>(h, A, b, st)));
      A.zb_mode = 0;
    } else {
      for (int P = 0; P <= 2 * m_t - 2; P++) {
        if (n_t % 2 == 0 && P == n_t - 1) continue;
        A.allcoef = P & 1;
        A.lin_base = (size_t)((P & 1) ? m_t + (P - 1) / 2 : P / 2) * sT;
        A.load = ds->load4 + (size_t)P * M;
        if (top) TRY((launch_loadvec<T, QTL>(h, A, b, st)));
        else TRY((launch_loadvec<T, QT>(h, A, b, st)));
      }
    }
    A.allcoef = 0;
    // ---- t-sweep, Thomas solves f, c, r, t; the last one subtracts from the coarse nodes
    {
      TRY((tsweep_launch<T>(h, ds->load4, ds->corr4, M, m_t, ds->nd[l].mass[0], st)));
    }
    const uint32_t m3a[3] = {(uint32_t)(m_t * Mc[1]), (uint32_t)Mc[2], (uint32_t)Mc[3]};
    TRY(ipk_launch<T>(h, 2, m3a, ds->corr4, ds->nd[l].thomas[3], nullptr, +1, st));
    TRY(ipk_launch<T>(h, 1, m3a, ds->corr4, ds->nd[l].thomas[2], nullptr, +1, st));
    TRY(ipk_launch<T>(h, 0, b.m, ds->corr4, ds->nd[l].thomas[1], nullptr, +1, st, (uint32_t)m_t, M));
    TRY((tsolve_apply<T>(h, ds->corr4, ds->nodal4[l - 1], M, m_t, Mc[1] * Mc[2], Mc[3], ds->nd[l].thomas[0], -1, st)));
    // ---- node restore, slice by slice
    T *fine = (l == L) ? data : ds->nodal4[l];
    const size_t fT = (l == L) ? full[0] : (size_t)N[1] * N[2] * N[3];
    A.fI = (l == L) ? full[1] : (size_t)N[2] * N[3];
    A.fJ = (l == L) ? full[2] : (size_t)N[3];
    if (batch) {
      A.zb_mt = m_t;
      A.zb_nt = n_t;
      A.zb_sT = sT;
      A.zb_M = M;
      A.zb_fT = fT;
      A.fine = fine;
      A.coarse = ds->nodal4[l - 1];
      A.zb_mode = 2;
      if (top) TRY((launch_restore<T, QTL, false>(h, A, b, "restore_q", st)));
      else TRY((launch_restore<T, QT, false>(h, A, b, "restore_q", st)));
      if (n_t - m_t > 0) {
        A.zb_mode = 3;
        if (top) TRY((launch_restore<T, QTL, true>(h, A, b, "restore_q_odd", st)));
        else TRY((launch_restore<T, QT, true>(h, A, b, "restore_q_odd", st)));
      }
      continue;
    }
    for (int tp = 0; tp < n_t; tp++) {
      const bool last_even = n_t % 2 == 0 && tp == n_t - 1;  // the real last node: coarse m_t - 1
      A.fine = fine + (size_t)tp * fT;
      if (!(tp & 1) || last_even) {
        const int zi = last_even ? m_t - 1 : tp / 2;
        A.coarse = ds->nodal4[l - 1] + (size_t)zi * M;
        A.lin_base = (size_t)zi * sT;
        if (top) TRY((launch_restore<T, QTL, false>(h, A, b, "restore_q", st)));
        else TRY((launch_restore<T, QT, false>(h, A, b, "restore_q", st)));
      } else {
        const int zi = (tp - 1) / 2;
        A.coarse = ds->nodal4[l - 1] + (size_t)zi * M;
        A.coarse_b = ds->nodal4[l - 1] + (size_t)(zi + 1) * M;
        A.tpos = tp;
        A.lin_base = (size_t)(m_t + zi) * sT;
        if (top) TRY((launch_restore<T, QTL, true>(h, A, b, "restore_q_odd", st)));
        else TRY((launch_restore<T, QT, true>(h, A, b, "restore_q_odd", st)));
      }
    }
  }
  return MGH_SUCCESS;
}

template <typename T>
int dequantize_recompose_fused(mgh_hierarchy *h, int64_t *q, int ebtype, double tol, double s,
                               double norm, uint64_t dict_size, int prep_huffman,
                               const uint64_t *oidx, const int64_t *oval, uint64_t ocount, T *data,
                               hipStream_t st) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const int L = h->L;
  if (prep_huffman && ocount) {
    TRY(launch(h, "outlier_restore", st, [&] {
      k_outlier_restore<<<(unsigned)((ocount + 255) / 256), 256, 0, st>>>(q, h->total, oidx, oval, ocount);
    }));
  }
  std::vector<T> qz(L + 1);
  hh->quantizers(ebtype, (T)tol, (T)s, (T)norm, false, qz.data());
  const bool calc_vol = !((T)s == std::numeric_limits<T>::infinity());
  RecomposeArgs<T> A{};
  A.q = q;
  A.dI = ds->full_I;
  A.dJ = ds->full_J;
  A.half = prep_huffman ? (int64_t)(dict_size / 2) : 0;
  std::vector<T> level_qv(L + 1);
  for (int l = 0; l <= L; l++) level_qv[l] = qz[l] * (calc_vol ? hh->level_volume(l, true) : (T)1);
  if (h->D == 4) return recompose_levels4<T, int64_t>(h, A, level_qv, data, st);
  return recompose_levels<T, int64_t>(h, A, level_qv, data, st);
}

// The same from 16-bit dictionary symbols (what the Huffman decoder of the high-level path
// delivers): no int64 array, the out-of-dictionary values are found through a hash table.
template <typename T>
int dequantize_recompose_fused16(mgh_hierarchy *h, const uint16_t *sym, int ebtype, double tol, double s,
                                 double norm, uint64_t dict_size, const uint64_t *oidx,
                                 const int64_t *oval, uint64_t ocount, T *data, hipStream_t st) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const int L = h->L;
  RecomposeArgs<T> A{};
  if (ocount) {
    size_t slots = 16;
    while (slots < 2 * ocount) slots *= 2;
    if (slots > ((size_t)1 << 31)) return fail(MGH_ERR_INVALID_ARGUMENT, "too many outliers");
    if (slots > ds->oh_slots) {
      (void)hipFree(ds->oh_key);
      (void)hipFree(ds->oh_val);
      ds->oh_key = nullptr;
      ds->oh_val = nullptr;
      ds->oh_slots = 0;
      HIP_TRY(hipMalloc(&ds->oh_key, slots * 8));
      HIP_TRY(hipMalloc(&ds->oh_val, slots * 8));
      ds->oh_slots = slots;
    }
    HIP_TRY(hipMemsetAsync(ds->oh_key, 0, slots * 8, st));
    TRY(launch(h, "outlier_table", st, [&] {
      k_outlier_hash_build<<<(unsigned)((ocount + 255) / 256), 256, 0, st>>>(
          oidx, oval, ocount, h->total, ds->oh_key, ds->oh_val, (uint32_t)(slots - 1));
    }));
    A.oh_key = ds->oh_key;
    A.oh_val = ds->oh_val;
    A.oh_mask = (uint32_t)(slots - 1);
  }
  std::vector<T> qz(L + 1);
  hh->quantizers(ebtype, (T)tol, (T)s, (T)norm, false, qz.data());
  const bool calc_vol = !((T)s == std::numeric_limits<T>::infinity());
  A.q16 = sym;
  A.dI = ds->full_I;
  A.dJ = ds->full_J;
  A.half = (int64_t)(dict_size / 2);
  std::vector<T> level_qv(L + 1);
  for (int l = 0; l <= L; l++) level_qv[l] = qz[l] * (calc_vol ? hh->level_volume(l, true) : (T)1);
  // Symbol width PER LEVEL: the finest levels -- where out-of-dictionary values are rare -- are
  // read as 16-bit symbols with the table look-up behind symbol 0. The levels below hold nearly
  // all the outliers, and a look-up there is a dependent global load inside latency-bound
  // kernels (measured: slower than the int64 path at 256^3). Their symbols -- the coarse corner
  // box of the reordered layout -- are widened to int64 in a compact box with the outliers
  // written over them, and those levels run the int64 kernels on it. Two levels stay on symbols
  // where the hierarchy is deep enough: the box is then 1/64 (D = 3) of the array instead of 1/8
  // (512^3: widening the 257^3 box cost 51 us, almost all of it the 136 MB of int64 stores).
  if (h->sym16_mixed && L >= 2 && h->total >= ((uint64_t)1 << 18)) {
    const int ntop = L >= 4 ? 2 : 1;
    const auto &Mc = hh->level_shape[L - ntop];
    const auto &N = hh->level_shape[L];
    BoxMap bm{};
    for (int k = 0; k < 4; k++) bm.m[k] = bm.n[k] = 1;
    for (int d = 0; d < h->D; d++) {
      bm.m[4 - h->D + d] = (uint32_t)Mc[d];
      bm.n[4 - h->D + d] = (uint32_t)N[d];
    }
    const size_t box = (size_t)bm.m[0] * bm.m[1] * bm.m[2] * bm.m[3];
    if (box > ds->qbox_elems) {
      (void)hipFree(ds->qbox);
      ds->qbox = nullptr;
      ds->qbox_elems = 0;
      HIP_TRY(hipMalloc(&ds->qbox, box * sizeof(int64_t)));
      ds->qbox_elems = box;
    }
    const size_t rows = box / bm.m[3];
    if (rows >= ((size_t)1 << 32)) return fail(MGH_ERR_INVALID_ARGUMENT, "coarse box too large");
    TRY(launch(h, "widen_box", st, [&] {
      k_widen_box<<<(unsigned)std::min<size_t>((rows + 3) / 4, 256 * 32), 256, 0, st>>>(sym, ds->qbox, bm, (uint32_t)rows);
    }));
    if (ocount)
      TRY(launch(h, "outlier_restore", st, [&] {
        k_outlier_restore_box<<<(unsigned)((ocount + 255) / 256), 256, 0, st>>>(ds->qbox, bm, oidx, oval, ocount);
      }));
    RecomposeArgs<T> A64{};
    A64.q = ds->qbox;
    A64.dJ = bm.m[3];
    A64.dI = (size_t)bm.m[2] * bm.m[3];
    A64.half = A.half;
    if (h->D == 4)
      return recompose_levels4<T, int64_t, uint16_t>(h, A64, level_qv, data, st, &A, (size_t)bm.m[1] * bm.m[2] * bm.m[3], ntop);
    return recompose_levels<T, int64_t, uint16_t>(h, A64, level_qv, data, st, &A, ntop);
  }
  if (h->D == 4) return recompose_levels4<T, uint16_t>(h, A, level_qv, data, st);
  return recompose_levels<T, uint16_t>(h, A, level_qv, data, st);
}

template <typename T>
int upload_quantizers(mgh_hierarchy *h, int ebtype, double tol, double s, double norm,
                      bool reciprocal, hipStream_t st) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const int L = h->L;
  std::vector<T> q(2 * (L + 1));
  hh->quantizers(ebtype, (T)tol, (T)s, (T)norm, reciprocal, q.data());
  const bool calc_vol = !((T)s == std::numeric_limits<T>::infinity());
  for (int l = 0; l <= L; l++) q[L + 1 + l] = calc_vol ? hh->level_volume(l, !reciprocal) : (T)1;
  ds->qmeta.calc_vol = calc_vol ? 1 : 0;
  // pageable-memory async copies are staged by the runtime before returning
  HIP_TRY(hipMemcpyAsync(ds->qz, q.data(), q.size() * sizeof(T), hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MGH_SUCCESS;
}

// the stand-alone quantizer with the table already in ds->qz (uploaded, or made on the device)
template <typename T>
int quantize_launch(mgh_hierarchy *h, const T *coeff, uint64_t dict_size, int prep_huffman, int64_t *q,
                    uint64_t *ocount, uint64_t *oidx, int64_t *oval, uint64_t ocap, hipStream_t st) {
  auto *ds = DS<T>(h);
  if (ocount) HIP_TRY(hipMemsetAsync(ocount, 0, sizeof(uint64_t), st));
  const size_t total = h->total;
  const unsigned grid = (unsigned)std::min<size_t>((total + kQuantPerRound - 1) / kQuantPerRound, 256 * 32);
  return launch(h, "quantize", st, [&] {
    k_quantize<T><<<grid, 256, 0, st>>>(ds->qmeta, total, coeff, ds->marks, ds->qz,
                                        ds->qz + (h->L + 1), (int64_t)dict_size, prep_huffman, q,
                                        (unsigned long long *)ocount, oidx, oval,
                                        (unsigned long long)ocap);
  });
}

template <typename T>
int quantize_impl(mgh_hierarchy *h, const T *coeff, int ebtype, double tol, double s, double norm,
                  uint64_t dict_size, int prep_huffman, int64_t *q, uint64_t *ocount,
                  uint64_t *oidx, int64_t *oval, uint64_t ocap, hipStream_t st) {
  TRY(upload_quantizers<T>(h, ebtype, tol, s, norm, true, st));
  return quantize_launch<T>(h, coeff, dict_size, prep_huffman, q, ocount, oidx, oval, ocap, st);
}

template <typename T>
int dequantize_impl(mgh_hierarchy *h, int64_t *q, int ebtype, double tol, double s, double norm,
                    uint64_t dict_size, int prep_huffman, const uint64_t *oidx,
                    const int64_t *oval, uint64_t ocount, T *coeff, hipStream_t st) {
  auto *ds = DS<T>(h);
  TRY(upload_quantizers<T>(h, ebtype, tol, s, norm, false, st));
  const size_t total = h->total;
  if (prep_huffman && ocount) {
    TRY(launch(h, "outlier_restore", st, [&] {
      k_outlier_restore<<<(unsigned)((ocount + 255) / 256), 256, 0, st>>>(q, h->total, oidx, oval, ocount);
    }));
  }
  const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
  TRY(launch(h, "dequantize", st, [&] {
    k_dequantize<T><<<grid, 256, 0, st>>>(ds->qmeta, total, q, ds->marks, ds->qz,
                                          ds->qz + (h->L + 1), (int64_t)dict_size, prep_huffman,
                                          coeff);
  }));
  return MGH_SUCCESS;
}

// ---- pitched caller arrays (mgh_set_ld; mgard_x::Array::ld, Array.hpp:70-84: hipMallocPitch pads the
// fastest dimension; SubArray.hpp:136-139 carries one ld per dimension) ------------------------
// Rows of the array (all dimensions but the fastest, right-aligned, leading 1s) and the element
// strides of those dimensions in the pitched array.
struct LdView {
  uint32_t ext[MGH_MAX_DIM];
  uint64_t stride[MGH_MAX_DIM];
  uint64_t rows;
};
inline LdView ld_view(const mgh_hierarchy *h, int which) {
  LdView V{};
  uint64_t st = 1, str[MGH_MAX_DIM] = {};
  for (int d = h->D - 1; d >= 0; d--) {
    str[d] = st;
    st *= h->has_ld[which] ? h->ld[which][d] : h->shape[d];
  }
  V.rows = 1;
  for (int k = 0; k < MGH_MAX_DIM; k++) {
    const int d = k - (MGH_MAX_DIM - h->D);
    V.ext[k] = d >= 0 ? (uint32_t)h->shape[d] : 1u;
    V.stride[k] = d >= 0 ? str[d] : 0;
    if (k < MGH_MAX_DIM - 1) V.rows *= V.ext[k];
  }
  return V;
}
__device__ __forceinline__ uint64_t ld_row_offset(const LdView &V, uint64_t row) {
  uint64_t r = row, off = 0;
#pragma unroll
  for (int d = MGH_MAX_DIM - 2; d >= 0; d--) {
    const uint64_t q = r / V.ext[d];
    off += (r - q * V.ext[d]) * V.stride[d];
    r = q;
  }
  return off;
}
// dense <-> pitched, one wave per row
template <typename T>
__global__ void __launch_bounds__(256) k_ld_copy(T *__restrict__ dense, T *__restrict__ pitched, LdView V, int to_dense) {
  const int lane = threadIdx.x & 63;
  const uint32_t nf = V.ext[MGH_MAX_DIM - 1];
  for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < V.rows; row += (uint64_t)gridDim.x * 4) {
    T *p = pitched + ld_row_offset(V, row);
    T *d = dense + row * nf;
    if (to_dense)
      for (uint32_t k = lane; k < nf; k += 64) d[k] = p[k];
    else
      for (uint32_t k = lane; k < nf; k += 64) p[k] = d[k];
  }
}
// the norm reductions over the rows of a pitched array (same accumulation into `out` as k_absmax / k_sqsum)
template <typename T, bool SQ>
__global__ void __launch_bounds__(256) k_norm_ld(const T *__restrict__ v, LdView V, unsigned long long *out) {
  const int lane = threadIdx.x & 63;
  const uint32_t nf = V.ext[MGH_MAX_DIM - 1];
  T acc = 0;
  for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < V.rows; row += (uint64_t)gridDim.x * 4) {
    const T *p = v + ld_row_offset(V, row);
    for (uint32_t k = lane; k < nf; k += 64) {
      const T x = p[k];
      if (SQ) {
        acc += x * x;
      } else {
        const T a = abs_t(x);
        acc = a > acc ? a : acc;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const T o = __shfl_down(acc, off, 64);
    if (SQ) acc += o; else acc = o > acc ? o : acc;
  }
  if (lane == 0) {
    if (SQ) {
      atomicAdd(reinterpret_cast<double *>(out), (double)acc);
    } else if (sizeof(T) == 4) {
      atomicMax(out, (unsigned long long)__float_as_uint((float)acc));
    } else {
      atomicMax(out, (unsigned long long)__double_as_longlong((double)acc));
    }
  }
}
inline unsigned ld_grid(const LdView &V) { return (unsigned)std::min<uint64_t>((V.rows + 3) / 4, 256 * 32); }

// The norm reduction of `data` (dense, or pitched with the strides of `view`) accumulated into `slot`.
template <typename T>
int norm_reduce(mgh_hierarchy *h, const T *data, double s, unsigned long long *slot, const LdView *view,
                size_t n_cold, hipStream_t st, unsigned long long *zero_a = nullptr, unsigned long long *zero_b = nullptr) {
  const bool inf = (T)s == std::numeric_limits<T>::infinity();
  if (view) {
    if (inf) return launch(h, "absmax", st, [&] { k_norm_ld<T, false><<<ld_grid(*view), 256, 0, st>>>(data, *view, slot); });
    return launch(h, "sqsum", st, [&] { k_norm_ld<T, true><<<ld_grid(*view), 256, 0, st>>>(data, *view, slot); });
  }
  const size_t total = h->total;
  const unsigned grid = (unsigned)std::min<size_t>((total + 1023) / 1024, 256 * 8);
  if (inf) return launch(h, "absmax", st, [&] { k_absmax<T><<<grid, 256, 0, st>>>(data, total, slot, n_cold, zero_a, zero_b); });
  return launch(h, "sqsum", st, [&] { k_sqsum<T><<<grid, 256, 0, st>>>(data, total, (double *)slot, n_cold, zero_a, zero_b); });
}

// Launch the norm reduction; the result stays in ds->scalar (absmax bits or double sum).
template <typename T> int norm_launch(mgh_hierarchy *h, const T *data, double s, hipStream_t st) {
  auto *ds = DS<T>(h);
  HIP_TRY(hipMemsetAsync(ds->scalar, 0, 8, st));
  if (h->has_ld[0] && !h->ld_guard) {
    const LdView V = ld_view(h, 0);
    return norm_reduce<T>(h, data, s, ds->scalar, &V, 0, st);
  }
  return norm_reduce<T>(h, data, s, ds->scalar, nullptr, 0, st);
}

// Quantizer table on the device from a device-resident norm (no host round trip).
template <typename T>
int fill_qparam_args(mgh_hierarchy *h, const T *d_norm, int ebtype, double tol, double s,
                     int decomposed, uint64_t nsub, uint64_t *reset_count, QParamArgs<T> &P) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  if (h->L + 1 > kMaxLevels) return fail(MGH_ERR_INVALID_ARGUMENT, "too many levels");
  P = QParamArgs<T>{};
  P.d_norm = d_norm;
  P.scalar = ds->scalar;
  P.s_is_inf = ((T)s == std::numeric_limits<T>::infinity()) ? 1 : 0;
  P.rel = ebtype == MGH_REL ? 1 : 0;
  P.decomposed = decomposed;
  P.normalize = hh->normalize_coordinates ? 1 : 0;
  P.total = h->total;
  P.nsub = nsub;
  P.tol = (T)tol;
  P.nlev = h->L + 1;
  hh->quantizer_denominators((T)s, P.den);
  for (int l = 0; l <= h->L; l++) P.vol[l] = P.s_is_inf ? (T)1 : hh->level_volume(l, false);
  P.qp = ds->qz;
  P.norm_out = ds->normval;
  P.reset_count = (unsigned long long *)reset_count;
  return MGH_SUCCESS;
}

template <typename T>
int make_qparams_launch(mgh_hierarchy *h, const T *d_norm, int ebtype, double tol, double s,
                        int decomposed, uint64_t nsub, uint64_t *reset_count, hipStream_t st) {
  QParamArgs<T> P;
  TRY(fill_qparam_args<T>(h, d_norm, ebtype, tol, s, decomposed, nsub, reset_count, P));
  return launch(h, "make_qparams", st, [&] { k_make_qparams<T><<<1, 64, 0, st>>>(P); });
}

template <typename T>
int norm_impl(mgh_hierarchy *h, const T *data, double s, double *out, hipStream_t st) {
  auto *ds = DS<T>(h);
  auto *hh = HH<T>(h);
  const size_t total = h->total;
  TRY(norm_launch<T>(h, data, s, st));
  T norm;
  if ((T)s == std::numeric_limits<T>::infinity()) {
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, ds->scalar, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sizeof(T) == 4) {
      uint32_t b32 = (uint32_t)bits;
      float f;
      std::memcpy(&f, &b32, 4);
      norm = (T)f;
    } else {
      double d;
      std::memcpy(&d, &bits, 8);
      norm = (T)d;
    }
  } else {
    double sum = 0;
    HIP_TRY(hipMemcpyAsync(&sum, ds->scalar, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    norm = (T)sum;
    // NormCalculator.hpp:62-66
    norm = hh->normalize_coordinates ? std::sqrt(norm / (T)total) : std::sqrt(norm);
  }
  if (norm == 0) norm = std::numeric_limits<T>::epsilon();
  *out = (double)norm;
  return MGH_SUCCESS;
}

template <typename T>
int64_t table_impl(const mgh_hierarchy *h, int kind, int level, int dim, void *out, uint64_t cap) {
  auto *hh = HH<T>(h);
  if (dim < 0 || dim >= hh->D) return fail(MGH_ERR_INVALID_ARGUMENT, "dim");
  if (kind == 4) {
    const auto &m = hh->marks[dim];
    if (cap < m.size()) return fail(MGH_ERR_INVALID_ARGUMENT, "capacity");
    std::memcpy(out, m.data(), m.size() * sizeof(int));
    return (int64_t)m.size();
  }
  if (level < 0 || level > hh->L) return fail(MGH_ERR_INVALID_ARGUMENT, "level");
  const auto &q = hh->lv[level][dim];
  const std::vector<T> *v = kind == 0 ? &q.dist : kind == 1 ? &q.ratio : kind == 2 ? &q.am
                            : kind == 3 ? &q.bm : nullptr;
  if (!v) return fail(MGH_ERR_INVALID_ARGUMENT, "kind");
  if (cap < v->size()) return fail(MGH_ERR_INVALID_ARGUMENT, "capacity");
  std::memcpy(out, v->data(), v->size() * sizeof(T));
  return (int64_t)v->size();
}

template <typename T>
int fused_q_entry(mgh_hierarchy *h, const T *data, int ebtype, double tol, double s, double norm,
                  uint64_t dict_size, int prep_huffman, int64_t *q, uint64_t *ocount,
                  uint64_t *oidx, int64_t *oval, uint64_t ocap, hipStream_t st) {
  QuantParams<T> qp = make_quant_params<T>(h, ebtype, tol, s, norm, true);
  qp.dict_size = (int64_t)dict_size;
  qp.prep_huffman = prep_huffman;
  qp.q = q;
  qp.ocount = (unsigned long long *)ocount;
  qp.oidx = oidx;
  qp.oval = oval;
  qp.ocap = ocap;
  return decompose_fused<T, OUT_Q>(h, data, nullptr, &qp, st);
}

// Same with the norm (and hence the quantizers) never leaving the device: d_norm given, or
// computed here (REL). h_norm_out != NULL costs one synchronisation at the END of the call.
template <typename T>
int fused_q_entry_device(mgh_hierarchy *h, const T *data, int ebtype, double tol, double s,
                         const T *d_norm, int decomposed, uint64_t nsub, double *h_norm_out,
                         uint64_t dict_size, int prep_huffman, int64_t *q, uint64_t *ocount,
                         uint64_t *oidx, int64_t *oval, uint64_t ocap, hipStream_t st,
                         uint16_t *q16 = nullptr) {
  auto *ds = DS<T>(h);
  // (mgh_norm_stream_*: the reduction is in the slot already; anything else the caller passes
  // alongside -- a given norm, an ABS bound -- overrides it)
  const bool streamed = ds->norm_streamed && !d_norm && ebtype == MGH_REL;
  ds->norm_streamed = false;
  const bool need_norm = !d_norm && ebtype == MGH_REL && !streamed;
  bool inline_qp = false;
  // The norm scalar has two slots used alternately: this call reduces into scalar[slot] (zero on
  // entry) and k_make_qparams zeroes the other one for the next call, together with the outlier
  // counter -- two memset launches less per step.
  if (ds->fscal_dirty && !streamed) HIP_TRY(hipMemsetAsync(ds->fscal, 0, 16, st));
  ds->fscal_dirty = true;
  unsigned long long *slot = ds->fscal + ds->scalar_slot;
  unsigned long long *other = ds->fscal + (1 - ds->scalar_slot);
  if (need_norm) {
    // all but the last MGH_ABSMAX_WARM_MB of the input with nontemporal loads: the level pass
    // re-reads the input from its end, and only what the norm pass read last can still be in
    // the 256 MB memory-side cache (512^3 f32, same box, 60 steps each: absmax 109 -> 93 us,
    // top-level pass 384 -> 397 us, step 0.894 -> 0.889 ms)
    const size_t total = h->total, warm = ((size_t)h->absmax_warm_mb << 20) / sizeof(T);
    if (ds->src_J) {  // (pitched input read in place: row by row)
      const LdView V = ld_view(h, 0);
      TRY(norm_reduce<T>(h, data, s, slot, &V, 0, st));
    } else {
      inline_qp = h->inline_qp && h->D == 3 && !decomposed;
      TRY(norm_reduce<T>(h, data, s, slot, nullptr, total > warm ? total - warm : 0, st,
                         inline_qp ? (unsigned long long *)ocount : nullptr, inline_qp ? other : nullptr));
    }
  }
  QuantParams<T> qp;
  if (inline_qp) {
    // the call's constants in device memory (uploaded when they change: per hierarchy they are a
    // function of bound type, tolerance and s)
    QParamArgs<T> P;
    TRY(fill_qparam_args<T>(h, d_norm, ebtype, tol, s, decomposed, nsub, nullptr, P));
    P.scalar = nullptr;  // (the slot alternates: it travels in FusedArgs::qslot)
    if (!ds->qinl_dev) TRY(dev_alloc(h, &ds->qinl_dev, (size_t)1));
    if (!ds->qinl_valid || std::memcmp(&ds->qinl_host, &P, sizeof(P)) != 0) {
      std::memcpy(&ds->qinl_host, &P, sizeof(P));
      HIP_TRY(hipMemcpyAsync(ds->qinl_dev, &ds->qinl_host, sizeof(P), hipMemcpyHostToDevice, st));
      ds->qinl_valid = true;
    }
    qp.d_qinl = ds->qinl_dev;
    qp.d_qslot = slot;
    qp.inline_done = [ds] {
      ds->scalar_slot = 1 - ds->scalar_slot;
      ds->fscal_dirty = false;
      return (int)MGH_SUCCESS;
    };
  }
  auto qparams = [&] {
    QParamArgs<T> P;
    TRY(fill_qparam_args<T>(h, d_norm, ebtype, tol, s, decomposed, nsub, ocount, P));
    P.scalar = slot;
    P.zero_next = other;
    TRY(launch(h, "make_qparams", st, [&] { k_make_qparams<T><<<1, 64, 0, st>>>(P); }));
    ds->scalar_slot = 1 - ds->scalar_slot;
    ds->fscal_dirty = false;
    return (int)MGH_SUCCESS;
  };
  qp.d_qp = ds->qz;
  qp.dict_size = (int64_t)dict_size;
  qp.prep_huffman = prep_huffman;
  qp.q = q;
  qp.q16 = q16;
  qp.ocount = (unsigned long long *)ocount;
  qp.oidx = oidx;
  qp.oval = oval;
  qp.ocap = ocap;
  TRY((decompose_fused<T, OUT_Q>(h, data, nullptr, &qp, st, qparams)));
  if (h_norm_out) {
    T nv = 0;
    HIP_TRY(hipMemcpyAsync(&nv, ds->normval, sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *h_norm_out = (double)nv;
  }
  return MGH_SUCCESS;
}

// Dense view of the caller's pitched arrays for the paths that do not take strides. ld_pack: the
// input copied into a dense buffer of the hierarchy (p then points there); LdOut: the kernels write
// a dense buffer, finish() spreads it into the caller's pitched array. Dense callers pay nothing.
template <typename T> int ld_pack(mgh_hierarchy *h, const T *&p, hipStream_t st) {
  if (!h->has_ld[0] || !p) return MGH_SUCCESS;
  auto *ds = DS<T>(h);
  if (!ds->pack_in) TRY(dev_alloc(h, &ds->pack_in, (size_t)h->total));
  const LdView V = ld_view(h, 0);
  TRY(launch(h, "ld_pack", st, [&] { k_ld_copy<T><<<ld_grid(V), 256, 0, st>>>(ds->pack_in, const_cast<T *>(p), V, 1); }));
  p = ds->pack_in;
  return MGH_SUCCESS;
}
template <typename T> struct LdOut {
  T *user = nullptr;
  int begin(mgh_hierarchy *h, T *&p) {
    if (!h->has_ld[1] || !p) return MGH_SUCCESS;
    auto *ds = DS<T>(h);
    if (!ds->pack_out) TRY(dev_alloc(h, &ds->pack_out, (size_t)h->total));
    user = p;
    p = ds->pack_out;
    return MGH_SUCCESS;
  }
  int finish(mgh_hierarchy *h, hipStream_t st) {
    if (!user) return MGH_SUCCESS;
    auto *ds = DS<T>(h);
    const LdView V = ld_view(h, 1);
    return launch(h, "ld_unpack", st, [&] { k_ld_copy<T><<<ld_grid(V), 256, 0, st>>>(ds->pack_out, user, V, 0); });
  }
};
// Strides of a pitched array in the 3-D view of the fused kernels, when they can read / write it in
// place: D <= 3 on the fused path, planes addressable in 32 bits like the dense ones.
inline bool ld_native3(const mgh_hierarchy *h, int which, size_t &sI, size_t &sJ);

// Norm accumulated over parts of the input (mgh_norm_stream_begin / _add): the same reduction
// kernels on a range, into the slot the next fused call reads.
template <typename T> int norm_stream_begin(mgh_hierarchy *h, hipStream_t st) {
  auto *ds = DS<T>(h);
  if (ds->fscal_dirty) HIP_TRY(hipMemsetAsync(ds->fscal, 0, 16, st));
  ds->fscal_dirty = true;  // (the slot is in use from here on)
  ds->norm_streamed = true;
  return MGH_SUCCESS;
}
template <typename T>
int norm_stream_add(mgh_hierarchy *h, const T *part, size_t count, double s, int cold, hipStream_t st) {
  auto *ds = DS<T>(h);
  if (!ds->norm_streamed) return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_norm_stream_add without mgh_norm_stream_begin");
  if (count == 0) return MGH_SUCCESS;
  unsigned long long *slot = ds->fscal + ds->scalar_slot;
  const unsigned grid = (unsigned)std::min<size_t>((count + 1023) / 1024, 256 * 8);
  if ((T)s == std::numeric_limits<T>::infinity())
    return launch(h, "absmax", st, [&] { k_absmax<T><<<grid, 256, 0, st>>>(part, count, slot, cold ? count : 0); });
  return launch(h, "sqsum", st, [&] { k_sqsum<T><<<grid, 256, 0, st>>>(part, count, (double *)slot, cold ? count : 0); });
}

#define DISPATCH(h, call_f, call_d)                                              \
  ((h)->dtype == MGH_FLOAT ? (call_f) : (call_d))

inline bool ld_native3(const mgh_hierarchy *h, int which, size_t &sI, size_t &sJ) {
  if (!h->has_ld[which] || !fused_ok(h) || h->force_v1) return false;
  const uint64_t lf = h->ld[which][2], lc = h->ld[which][1];
  if (lf * lc >= ((uint64_t)1 << 30)) return false;
  sJ = (size_t)lf;
  sI = (size_t)(lf * lc);
  return true;
}

// An entry point with a pitched T input and / or output (mgh_set_ld): `call(in, out)` is the entry
// point itself, run once more under the guard with pointers it can take as dense ones -- copies
// (ld_pack / LdOut), or the caller's own arrays where the fused 3-D kernels take the strides
// (native_ok: the call is one that runs them).
template <typename T, typename F>
int ld_entry_t(mgh_hierarchy *h, const void *in, void *out, bool native_ok, hipStream_t st, F &&call) {
  auto *ds = DS<T>(h);
  const T *pin = (const T *)in;
  T *pout = (T *)out;
  size_t sI = 0, sJ = 0, oI = 0, oJ = 0;
  const bool nat_in = pin && native_ok && ld_native3(h, 0, sI, sJ);
  const bool nat_out = pout && native_ok && ld_native3(h, 1, oI, oJ);
  if (pin && h->has_ld[0] && !nat_in) TRY(ld_pack<T>(h, pin, st));
  LdOut<T> o;
  if (pout && h->has_ld[1] && !nat_out) TRY(o.begin(h, pout));
  if (nat_in) ds->src_I = sI, ds->src_J = sJ;
  if (nat_out) ds->dst_I = oI, ds->dst_J = oJ;
  h->ld_guard = true;
  const int rc = call((const void *)pin, (void *)pout);
  h->ld_guard = false;
  ds->src_I = ds->src_J = ds->dst_I = ds->dst_J = 0;
  if (rc != MGH_SUCCESS) return rc;
  return o.finish(h, st);
}
template <typename F>
int ld_entry(mgh_hierarchy *h, const void *in, void *out, bool native_ok, void *stream, F &&call) {
  if (h->dtype == MGH_FLOAT) return ld_entry_t<float>(h, in, out, native_ok, (hipStream_t)stream, call);
  return ld_entry_t<double>(h, in, out, native_ok, (hipStream_t)stream, call);
}
inline bool ld_wanted(const mgh_hierarchy *h, const void *in, const void *out) {
  return !h->ld_guard && ((in && h->has_ld[0]) || (out && h->has_ld[1]));
}

} // namespace

// REL bound without a norm on the shapes the fused level kernels do not take (D = 5, thin boxes ...):
// norm, quantizer table, decomposition and quantizer queued one behind the other -- the norm and the
// table stay on the device as in fused_q_entry_device (mgh_norm + mgh_quantize were two host round
// trips, 35 us each on the 5-D step). h_norm_out != NULL: one synchronisation at the END.
template <typename T>
int staged_q_entry_device(mgh_hierarchy *h, const T *data, int ebtype, double tol, double s, double *h_norm_out,
                          uint64_t dict_size, int prep_huffman, int64_t *q, uint64_t *ocount, uint64_t *oidx,
                          int64_t *oval, uint64_t ocap, T *coeff, hipStream_t st) {
  auto *ds = DS<T>(h);
  TRY(norm_launch<T>(h, data, s, st));
  TRY(make_qparams_launch<T>(h, nullptr, ebtype, tol, s, 0, 1, nullptr, st));
  ds->qmeta.calc_vol = ((T)s == std::numeric_limits<T>::infinity()) ? 0 : 1;
  TRY(mgh_decompose(h, data, coeff, (void *)st));
  TRY(quantize_launch<T>(h, coeff, dict_size, prep_huffman, q, ocount, oidx, oval, ocap, st));
  if (h_norm_out) {
    T nv;
    HIP_TRY(hipMemcpyAsync(&nv, ds->normval, sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *h_norm_out = (double)nv;
  }
  return MGH_SUCCESS;
}

extern "C" {

const char *mgh_last_error(void) { return g_last_error.c_str(); }

/* (internal: lets the high-level translation unit report through the same channel) */
void mgh_set_last_error_(const char *msg) { g_last_error = msg ? msg : ""; }

int mgh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mgh_hierarchy_create(mgh_hierarchy **out, int D, const uint64_t *shape, int dtype,
                         const void *const *h_coords, int normalize_coordinates,
                         uint64_t max_level, int device) {
  if (!out || !shape) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (D < 1 || D > MGH_MAX_DIM) return fail(MGH_ERR_UNSUPPORTED_DIMENSION, "D must be 1..5");
  if (dtype != MGH_FLOAT && dtype != MGH_DOUBLE) return fail(MGH_ERR_UNSUPPORTED_DTYPE, "dtype");
  if (mgh_device_count() <= device || device < 0)
    return fail(MGH_ERR_NO_DEVICE, "HIP device " + std::to_string(device) + " not available");
  HIP_TRY(hipSetDevice(device));
  {
    const std::string bad = env_validate();
    if (!bad.empty()) return fail(MGH_ERR_INVALID_ARGUMENT, bad);
  }
  auto *h = new mgh_hierarchy();
  {
    h->force_v1_env = (int)env_get("MGH_FORCE_V1", -1);
    h->force_v1 = h->force_v1_env == 1;
    h->force_nd = env_get("MGH_FORCE_ND", 0) != 0;
    h->force_nd_ipk = env_get("MGH_ND_IPK", 0) != 0;
    h->ipk_stream = (int)env_get("MGH_IPK_STREAM", h->ipk_stream);
    h->ipk_dma = (int)env_get("MGH_IPK_DMA", h->ipk_dma);
    h->ipk_dma_min_env = env_get("MGH_IPK_DMA_MIN", -1);
    h->absmax_warm_mb = (int)env_get("MGH_ABSMAX_WARM_MB", h->absmax_warm_mb);
    h->fused_faces = (int)env_get("MGH_FUSED_FACES", h->fused_faces);
    h->fused_xcd = (int)env_get("MGH_FUSED_XCD", h->fused_xcd);
    h->slice_batch = (int)env_get("MGH_SLICE_BATCH", h->slice_batch);
    h->fused_tall = (int)env_get("MGH_FUSED_TALL", h->fused_tall);
    h->fused_fixed = (int)env_get("MGH_FUSED_FIXED", h->fused_fixed);
    h->fused_wide = (int)env_get("MGH_FUSED_WIDE", -1);  // (-1: by data type, below)
    h->fused4 = (int)env_get("MGH_FUSED4", h->fused4);
    h->box = (int)env_get("MGH_BOX", h->box);
    h->restore_v = (int)env_get("MGH_RESTORE_V", h->restore_v);
    h->sym16_mixed = (int)env_get("MGH_SYM16_MIXED", h->sym16_mixed);
    h->ipk_spec = (int)env_get("MGH_IPK_SPEC", h->ipk_spec);
    h->ipk_spec_k = (int)env_get("MGH_IPK_SPEC_K", h->ipk_spec_k);
    h->ipk_spec_long = (int)env_get("MGH_IPK_SPEC_LONG", h->ipk_spec_long);
    h->ipk_spec_max = (uint32_t)env_get("MGH_IPK_SPEC_MAX", (long)h->ipk_spec_max);
    h->ipk_chunk = (int)env_get("MGH_IPK_CHUNK", h->ipk_chunk);
    h->outlier_agg = (int)env_get("MGH_OUTLIER_AGG", h->outlier_agg);
    h->ipk_chunk_k = (int)env_get("MGH_IPK_CHUNK_K", h->ipk_chunk_k);
    h->tail_solves = (int)env_get("MGH_TAIL_SOLVES", h->tail_solves);
    h->ipk_dma_rounds = (int)env_get("MGH_IPK_DMA_ROUNDS", h->ipk_dma_rounds);
    h->inline_qp = (int)env_get("MGH_INLINE_QP", h->inline_qp);
    h->nd_rows = (int)env_get("MGH_ND_ROWS", h->nd_rows);
    h->cls1 = (size_t)env_get("MGH_CLS1", (long)h->cls1);
    h->cls2 = (size_t)env_get("MGH_CLS2", (long)h->cls2);
    if (const char *e = std::getenv("MGH_RCH")) std::sscanf(e, "%d,%d,%d", &h->rch[0], &h->rch[1], &h->rch[2]);
    h->ipk_w = (uint32_t)env_get("MGH_IPK_W", h->ipk_w);
    h->ipk_pd = (int)env_get("MGH_IPK_PD", h->ipk_pd);
    h->ipk_wpc = (size_t)env_get("MGH_IPK_WPC", (long)h->ipk_wpc);
    h->ipk_kr16 = (int)env_get("MGH_IPK_KR16", h->ipk_kr16);
    h->ipk_range_mb = (int)env_get("MGH_IPK_RANGE_MB", h->ipk_range_mb);
    h->ipk_contig_rounds = (size_t)env_get("MGH_IPK_CONTIG", (long)h->ipk_contig_rounds);
    h->no_head = env_get("MGH_NO_RECOMPOSE_HEAD", 0) != 0;
    h->restore_rows = env_get("MGH_RESTORE_ROWS", 0) != 0;
    h->debug_sync = env_get("MGH_DEBUG_SYNC", 0) != 0;
    if (h->force_nd) h->force_v1 = true;  // keeps the fused entry points off
  }
  h->dtype = dtype;
  h->device = device;
  // tile shape of the long marches: 4 x 64 for floats, 8 x 32 for doubles -- fine rows of ~512 bytes
  // either way (512^3 f64 non-uniform, three alternating runs on one box: top-level pass
  // 612 -> 593 us, step 1.413 -> 1.384 ms)
  if (h->fused_wide < 0) h->fused_wide = dtype == MGH_FLOAT ? 1 : 0;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      h->num_cu = (size_t)cus;
  }
  h->ipk_dma_min = h->ipk_dma_min_env >= 0 ? (size_t)h->ipk_dma_min_env : 2 * h->num_cu;
  h->D = D;
  bool ok;
  if (dtype == MGH_FLOAT) {
    auto *hh = new HostHierarchy<float>();
    h->host = hh;
    ok = hh->init(D, shape, (const float *const *)h_coords, normalize_coordinates != 0, max_level);
    hh->normalize_coordinates = normalize_coordinates != 0;
    h->L = hh->L;
    h->total = ok ? hh->total() : 0;
  } else {
    auto *hh = new HostHierarchy<double>();
    h->host = hh;
    ok = hh->init(D, shape, (const double *const *)h_coords, normalize_coordinates != 0, max_level);
    hh->normalize_coordinates = normalize_coordinates != 0;
    h->L = hh->L;
    h->total = ok ? hh->total() : 0;
  }
  if (!ok) {
    if (dtype == MGH_FLOAT) delete HH<float>(h); else delete HH<double>(h);
    delete h;
    return fail(MGH_ERR_INVALID_ARGUMENT,
                "invalid shape: every dimension must have at least 3 nodes");
  }
  h->plane_elems = shape[D - 1] * (D >= 2 ? shape[D - 2] : 1);
  for (int d = 0; d < D; d++) h->shape[d] = shape[d];
  // Thin arrays (a long slowest dimension over planes of a few nodes: 1000000 x 5 x 5): the tiled
  // level kernels cover the coarse (c, f) plane with tiles of 8 x 32 nodes and march along r; where
  // the plane fills less than an eighth of its tiles the one-thread-per-element kernels are the
  // faster ones (4194304 x 3 x 3: 82 -> 12.5 ms per mgh_compress; step of 1000000 x 5 x 5 5.0 vs 6.2 ms,
  // of 100000 x 9 x 9 1.15 vs 1.30 ms -- and 300000 x 17 x 17, 16 % of its tiles, 5.6 vs 3.9 ms the
  // other way round since the tiles of a small cross-section reach all XCDs: round 6).
  if (h->force_v1_env < 0 && D == 3) {
    const uint64_t m1 = shape[1] / 2 + 1, m2 = shape[2] / 2 + 1;
    double tiles = (double)((m1 + 7) / 8) * (double)((m2 + 31) / 32);
    // (a short fastest extent under a long middle one: the 64 x 4 tiles -- fused_tall_tiles)
    if (h->fused_tall && m2 <= 16 && m1 >= 48) tiles = (double)((m1 + 63) / 64) * (double)((m2 + 3) / 4);
    if ((double)(m1 * m2) < 0.125 * tiles * 256.0) h->force_v1 = true;
  }
  int rc = DISPATCH(h, build_device_state<float>(h), build_device_state<double>(h));
  if (rc != MGH_SUCCESS) {
    mgh_hierarchy_destroy(h);
    return rc;
  }
  *out = h;
  return MGH_SUCCESS;
}

void mgh_hierarchy_destroy(mgh_hierarchy *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (auto &kv : h->prof)
    for (auto &ev : kv.second.pending) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  if (h->dtype == MGH_FLOAT) destroy_state<float>(h); else destroy_state<double>(h);
  delete h;
}

int mgh_l_target(const mgh_hierarchy *h) { return h ? h->L : MGH_ERR_INVALID_ARGUMENT; }

int mgh_level_shape(const mgh_hierarchy *h, int level, uint64_t *out_shape) {
  if (!h || !out_shape || level < 0 || level > h->L) return fail(MGH_ERR_INVALID_ARGUMENT, "level");
  for (int d = 0; d < h->D; d++)
    out_shape[d] = h->dtype == MGH_FLOAT ? HH<float>(h)->level_shape[level][d]
                                         : HH<double>(h)->level_shape[level][d];
  return MGH_SUCCESS;
}

uint64_t mgh_total_num_elems(const mgh_hierarchy *h) { return h ? h->total : 0; }
size_t mgh_device_bytes(const mgh_hierarchy *h) { return h ? h->device_bytes : 0; }
const void *mgh_norm_device_ptr(const mgh_hierarchy *h) {
  if (!h) return nullptr;
  return h->dtype == MGH_FLOAT ? (const void *)DS<float>(h)->normval : (const void *)DS<double>(h)->normval;
}


int64_t mgh_hierarchy_table(const mgh_hierarchy *h, int kind, int level, int dim, void *h_out,
                            uint64_t cap) {
  if (!h || !h_out) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  return DISPATCH(h, table_impl<float>(h, kind, level, dim, h_out, cap),
                  table_impl<double>(h, kind, level, dim, h_out, cap));
}

int mgh_set_ld(mgh_hierarchy *h, int which, const uint64_t *ld) {
  if (!h || (which != MGH_LD_IN && which != MGH_LD_OUT)) return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_set_ld: which");
  if (!ld) {
    h->has_ld[which] = false;
    return MGH_SUCCESS;
  }
  bool dense = true;
  for (int d = 1; d < h->D; d++) {
    if (ld[d] < h->shape[d]) return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_set_ld: a leading dimension is smaller than the extent");
    dense &= ld[d] == h->shape[d];
  }
  for (int d = 0; d < h->D; d++) h->ld[which][d] = d == 0 ? h->shape[0] : ld[d];
  h->has_ld[which] = !dense;
  return MGH_SUCCESS;
}

int mgh_norm(mgh_hierarchy *h, const void *d_data, double s, double *h_norm_out, void *stream) {
  if (!h || !d_data || !h_norm_out) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  return DISPATCH(h, norm_impl<float>(h, (const float *)d_data, s, h_norm_out, (hipStream_t)stream),
                  norm_impl<double>(h, (const double *)d_data, s, h_norm_out, (hipStream_t)stream));
}

int mgh_decompose(mgh_hierarchy *h, const void *d_data, void *d_coeff, void *stream) {
  if (!h || !d_data || !d_coeff) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  if (ld_wanted(h, d_data, d_coeff))
    return ld_entry(h, d_data, d_coeff, false, stream, [&](const void *i, void *o) { return mgh_decompose(h, i, o, stream); });
  return DISPATCH(h, decompose_impl<float>(h, (const float *)d_data, (float *)d_coeff, (hipStream_t)stream),
                  decompose_impl<double>(h, (const double *)d_data, (double *)d_coeff, (hipStream_t)stream));
}

int mgh_recompose(mgh_hierarchy *h, const void *d_coeff, void *d_data, void *stream) {
  if (!h || !d_data || !d_coeff) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  if (ld_wanted(h, d_coeff, d_data))
    return ld_entry(h, d_coeff, d_data, false, stream, [&](const void *i, void *o) { return mgh_recompose(h, i, o, stream); });
  return DISPATCH(h, recompose_impl<float>(h, (const float *)d_coeff, (float *)d_data, (hipStream_t)stream),
                  recompose_impl<double>(h, (const double *)d_coeff, (double *)d_data, (hipStream_t)stream));
}

int mgh_quantize(mgh_hierarchy *h, const void *d_coeff, int ebtype, double tol, double s,
                 double norm, uint64_t dict_size, int prep_huffman, int64_t *d_quantized,
                 uint64_t *d_outlier_count, uint64_t *d_outlier_idx, int64_t *d_outlier_val,
                 uint64_t outlier_capacity, void *stream) {
  if (!h || !d_coeff || !d_quantized) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (prep_huffman && (!d_outlier_count || (outlier_capacity && (!d_outlier_idx || !d_outlier_val))))
    return fail(MGH_ERR_INVALID_ARGUMENT, "outlier buffers required with prep_huffman");
  HIP_TRY(hipSetDevice(h->device));
  if (ld_wanted(h, d_coeff, nullptr))  // (pitched coefficients: quantized from a dense copy; the integers are always dense)
    return ld_entry(h, d_coeff, nullptr, false, stream, [&](const void *i, void *) {
      return mgh_quantize(h, i, ebtype, tol, s, norm, dict_size, prep_huffman, d_quantized, d_outlier_count,
                          d_outlier_idx, d_outlier_val, outlier_capacity, stream);
    });
  return DISPATCH(h,
                  quantize_impl<float>(h, (const float *)d_coeff, ebtype, tol, s, norm, dict_size,
                                       prep_huffman, d_quantized, d_outlier_count, d_outlier_idx,
                                       d_outlier_val, outlier_capacity, (hipStream_t)stream),
                  quantize_impl<double>(h, (const double *)d_coeff, ebtype, tol, s, norm, dict_size,
                                        prep_huffman, d_quantized, d_outlier_count, d_outlier_idx,
                                        d_outlier_val, outlier_capacity, (hipStream_t)stream));
}

int mgh_dequantize(mgh_hierarchy *h, int64_t *d_quantized, int ebtype, double tol, double s,
                   double norm, uint64_t dict_size, int prep_huffman,
                   const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                   uint64_t outlier_count, void *d_coeff, void *stream) {
  if (!h || !d_coeff || !d_quantized) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  if (ld_wanted(h, nullptr, d_coeff))  // (pitched coefficients out: dequantized densely, then spread)
    return ld_entry(h, nullptr, d_coeff, false, stream, [&](const void *, void *o) {
      return mgh_dequantize(h, d_quantized, ebtype, tol, s, norm, dict_size, prep_huffman, d_outlier_idx,
                            d_outlier_val, outlier_count, o, stream);
    });
  return DISPATCH(h,
                  dequantize_impl<float>(h, d_quantized, ebtype, tol, s, norm, dict_size,
                                         prep_huffman, d_outlier_idx, d_outlier_val, outlier_count,
                                         (float *)d_coeff, (hipStream_t)stream),
                  dequantize_impl<double>(h, d_quantized, ebtype, tol, s, norm, dict_size,
                                          prep_huffman, d_outlier_idx, d_outlier_val, outlier_count,
                                          (double *)d_coeff, (hipStream_t)stream));
}

int mgh_decompose_quantize_sym16(mgh_hierarchy *h, const void *d_data, int error_bound_type, double tol,
                                 double s, double norm, double *h_norm_out, uint64_t dict_size,
                                 uint16_t *d_symbols, uint64_t *d_outlier_count,
                                 uint64_t *d_outlier_idx, int64_t *d_outlier_val,
                                 uint64_t outlier_capacity, void *stream) {
  if (!h || !d_data || !d_symbols || !d_outlier_count || (outlier_capacity && (!d_outlier_idx || !d_outlier_val)))
    return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (dict_size == 0 || dict_size > 65536) return fail(MGH_ERR_INVALID_ARGUMENT, "dict_size must be in 1..65536");
  HIP_TRY(hipSetDevice(h->device));
  if (!(fusedc_ok(h) && !h->force_v1))
    return fail(MGH_ERR_UNSUPPORTED_DIMENSION, "16-bit symbols: only on the fused 3-D / 4-D path");
  if (ld_wanted(h, d_data, nullptr))
    return ld_entry(h, d_data, nullptr, true, stream, [&](const void *i, void *) {
      return mgh_decompose_quantize_sym16(h, i, error_bound_type, tol, s, norm, h_norm_out, dict_size, d_symbols,
                                          d_outlier_count, d_outlier_idx, d_outlier_val, outlier_capacity, stream);
    });
  // (the norm and the quantizers stay on the device; a given norm is uploaded first)
  const void *d_norm = nullptr;
  if (!(error_bound_type == MGH_REL && !(norm > 0)) && error_bound_type == MGH_REL) {
    if (h->dtype == MGH_FLOAT) {
      const float nv = (float)norm;
      HIP_TRY(hipMemcpyAsync(DS<float>(h)->normval, &nv, sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      d_norm = DS<float>(h)->normval;
    } else {
      const double nv = norm;
      HIP_TRY(hipMemcpyAsync(DS<double>(h)->normval, &nv, sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      d_norm = DS<double>(h)->normval;
    }
    if (h_norm_out) *h_norm_out = norm;
    h_norm_out = nullptr;
  }
  if (error_bound_type != MGH_REL) {  // (no norm involved)
    if (h_norm_out) *h_norm_out = norm;
    h_norm_out = nullptr;
  }
  return DISPATCH(h,
                  fused_q_entry_device<float>(h, (const float *)d_data, error_bound_type, tol, s,
                                              (const float *)d_norm, 0, 1, h_norm_out, dict_size, 1, nullptr,
                                              d_outlier_count, d_outlier_idx, d_outlier_val,
                                              outlier_capacity, (hipStream_t)stream, d_symbols),
                  fused_q_entry_device<double>(h, (const double *)d_data, error_bound_type, tol, s,
                                               (const double *)d_norm, 0, 1, h_norm_out, dict_size, 1, nullptr,
                                               d_outlier_count, d_outlier_idx, d_outlier_val,
                                               outlier_capacity, (hipStream_t)stream, d_symbols));
}

int mgh_sym16_supported(const mgh_hierarchy *h) {
  return h && fusedc_ok(h) && !h->force_v1 ? 1 : 0;
}

int mgh_dequantize_recompose_sym16(mgh_hierarchy *h, const uint16_t *d_symbols, int error_bound_type,
                                   double tol, double s, double norm, uint64_t dict_size,
                                   const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                                   uint64_t outlier_count, void *d_data_out, void *stream) {
  if (!h || !d_symbols || !d_data_out || (outlier_count && (!d_outlier_idx || !d_outlier_val)))
    return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (dict_size == 0 || dict_size > 65536) return fail(MGH_ERR_INVALID_ARGUMENT, "dict_size must be in 1..65536");
  HIP_TRY(hipSetDevice(h->device));
  if (!mgh_sym16_supported(h))
    return fail(MGH_ERR_UNSUPPORTED_DIMENSION, "16-bit symbols: only on the fused 3-D / 4-D path");
  if (ld_wanted(h, nullptr, d_data_out))
    return ld_entry(h, nullptr, d_data_out, true, stream, [&](const void *, void *o) {
      return mgh_dequantize_recompose_sym16(h, d_symbols, error_bound_type, tol, s, norm, dict_size, d_outlier_idx,
                                            d_outlier_val, outlier_count, o, stream);
    });
  return DISPATCH(h,
                  dequantize_recompose_fused16<float>(h, d_symbols, error_bound_type, tol, s, norm, dict_size,
                                                      d_outlier_idx, d_outlier_val, outlier_count,
                                                      (float *)d_data_out, (hipStream_t)stream),
                  dequantize_recompose_fused16<double>(h, d_symbols, error_bound_type, tol, s, norm, dict_size,
                                                       d_outlier_idx, d_outlier_val, outlier_count,
                                                       (double *)d_data_out, (hipStream_t)stream));
}

int mgh_decompose_quantize(mgh_hierarchy *h, const void *d_data, int error_bound_type, double tol, double s,
                           double norm, double *h_norm_out, uint64_t dict_size, int prep_huffman,
                           int64_t *d_quantized, uint64_t *d_outlier_count,
                           uint64_t *d_outlier_idx, int64_t *d_outlier_val,
                           uint64_t outlier_capacity, void *d_coeff_opt, void *stream) {
  if (!h || !d_data || !d_quantized) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  // (the fused level kernels test the dictionary range in 32 bits: larger dictionaries are staged)
  const bool fused = !d_coeff_opt && fusedc_ok(h) && !h->force_v1 && dict_size <= ((uint64_t)1 << 30);
  if (ld_wanted(h, d_data, d_coeff_opt))
    return ld_entry(h, d_data, d_coeff_opt, fused, stream, [&](const void *i, void *o) {
      return mgh_decompose_quantize(h, i, error_bound_type, tol, s, norm, h_norm_out, dict_size, prep_huffman,
                                    d_quantized, d_outlier_count, d_outlier_idx, d_outlier_val, outlier_capacity, o,
                                    stream);
    });
  if (fused && error_bound_type == MGH_REL && !(norm > 0)) {
    // the norm and the quantizers stay on the device: no host round trip inside the call
    if (prep_huffman && (!d_outlier_count || (outlier_capacity && (!d_outlier_idx || !d_outlier_val))))
      return fail(MGH_ERR_INVALID_ARGUMENT, "outlier buffers required with prep_huffman");
    return DISPATCH(h,
                    fused_q_entry_device<float>(h, (const float *)d_data, error_bound_type, tol, s,
                                                nullptr, 0, 1, h_norm_out, dict_size, prep_huffman,
                                                d_quantized, d_outlier_count, d_outlier_idx,
                                                d_outlier_val, outlier_capacity, (hipStream_t)stream),
                    fused_q_entry_device<double>(h, (const double *)d_data, error_bound_type, tol, s,
                                                 nullptr, 0, 1, h_norm_out, dict_size, prep_huffman,
                                                 d_quantized, d_outlier_count, d_outlier_idx,
                                                 d_outlier_val, outlier_capacity, (hipStream_t)stream));
  }
  if (!fused && error_bound_type == MGH_REL && !(norm > 0)) {
    void *coeff = d_coeff_opt;
    if (!coeff) {
      int rc = DISPATCH(h, ensure_scratch<float>(h), ensure_scratch<double>(h));
      if (rc != MGH_SUCCESS) return rc;
      coeff = h->dtype == MGH_FLOAT ? (void *)DS<float>(h)->scratch_full : (void *)DS<double>(h)->scratch_full;
      if (coeff == d_data) return fail(MGH_ERR_INVALID_ARGUMENT, "aliasing");
    }
    return DISPATCH(h,
                    staged_q_entry_device<float>(h, (const float *)d_data, error_bound_type, tol, s, h_norm_out,
                                                 dict_size, prep_huffman, d_quantized, d_outlier_count, d_outlier_idx,
                                                 d_outlier_val, outlier_capacity, (float *)coeff, (hipStream_t)stream),
                    staged_q_entry_device<double>(h, (const double *)d_data, error_bound_type, tol, s, h_norm_out,
                                                  dict_size, prep_huffman, d_quantized, d_outlier_count, d_outlier_idx,
                                                  d_outlier_val, outlier_capacity, (double *)coeff, (hipStream_t)stream));
  }
  if (error_bound_type == MGH_REL && !(norm > 0)) {
    int rc = mgh_norm(h, d_data, s, &norm, stream);
    if (rc != MGH_SUCCESS) return rc;
  }
  if (h_norm_out) *h_norm_out = norm;
  if (fused) {
    if (prep_huffman && (!d_outlier_count || (outlier_capacity && (!d_outlier_idx || !d_outlier_val))))
      return fail(MGH_ERR_INVALID_ARGUMENT, "outlier buffers required with prep_huffman");
    if (d_outlier_count) HIP_TRY(hipMemsetAsync(d_outlier_count, 0, sizeof(uint64_t), (hipStream_t)stream));
    return DISPATCH(h,
                    fused_q_entry<float>(h, (const float *)d_data, error_bound_type, tol, s, norm,
                                         dict_size, prep_huffman, d_quantized, d_outlier_count,
                                         d_outlier_idx, d_outlier_val, outlier_capacity,
                                         (hipStream_t)stream),
                    fused_q_entry<double>(h, (const double *)d_data, error_bound_type, tol, s, norm,
                                          dict_size, prep_huffman, d_quantized, d_outlier_count,
                                          d_outlier_idx, d_outlier_val, outlier_capacity,
                                          (hipStream_t)stream));
  }
  void *coeff = d_coeff_opt;
  if (!coeff) {
    int rc = DISPATCH(h, ensure_scratch<float>(h), ensure_scratch<double>(h));
    if (rc != MGH_SUCCESS) return rc;
    coeff = h->dtype == MGH_FLOAT ? (void *)DS<float>(h)->scratch_full
                                  : (void *)DS<double>(h)->scratch_full;
    if (coeff == d_data) return fail(MGH_ERR_INVALID_ARGUMENT, "aliasing");
  }
  int rc = mgh_decompose(h, d_data, coeff, stream);
  if (rc != MGH_SUCCESS) return rc;
  return mgh_quantize(h, coeff, error_bound_type, tol, s, norm, dict_size, prep_huffman,
                      d_quantized, d_outlier_count, d_outlier_idx, d_outlier_val,
                      outlier_capacity, stream);
}

int mgh_norm_device(mgh_hierarchy *h, const void *d_data, double s, void *d_norm_out,
                    void *stream) {
  if (!h || !d_data || !d_norm_out) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  if (h->dtype == MGH_FLOAT) {
    TRY(norm_launch<float>(h, (const float *)d_data, s, st));
    // ABS/undecomposed parameters are irrelevant here: only the norm conversion is wanted
    TRY(make_qparams_launch<float>(h, nullptr, MGH_ABS, 1.0, s, 0, 1, nullptr, st));
    HIP_TRY(hipMemcpyAsync(d_norm_out, DS<float>(h)->normval, sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    TRY(norm_launch<double>(h, (const double *)d_data, s, st));
    TRY(make_qparams_launch<double>(h, nullptr, MGH_ABS, 1.0, s, 0, 1, nullptr, st));
    HIP_TRY(hipMemcpyAsync(d_norm_out, DS<double>(h)->normval, sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  return MGH_SUCCESS;
}

int mgh_norm_stream_begin(mgh_hierarchy *h, void *stream) {
  if (!h) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (!fusedc_ok(h) || h->force_v1)
    return fail(MGH_ERR_UNSUPPORTED_DIMENSION, "streamed norm: only in front of the fused 3-D / 4-D path");
  HIP_TRY(hipSetDevice(h->device));
  return DISPATCH(h, norm_stream_begin<float>(h, (hipStream_t)stream), norm_stream_begin<double>(h, (hipStream_t)stream));
}

int mgh_norm_stream_add(mgh_hierarchy *h, const void *d_part, uint64_t count, double s, int cold, void *stream) {
  if (!h || (!d_part && count)) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (h->has_ld[0]) return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_norm_stream_add takes parts of a dense array (mgh_set_ld is set)");
  HIP_TRY(hipSetDevice(h->device));
  return DISPATCH(h, norm_stream_add<float>(h, (const float *)d_part, count, s, cold, (hipStream_t)stream),
                  norm_stream_add<double>(h, (const double *)d_part, count, s, cold, (hipStream_t)stream));
}

int mgh_decompose_quantize_dn(mgh_hierarchy *h, const void *d_data, int error_bound_type,
                              double tol, double s, const void *d_norm, uint64_t num_subdomains,
                              uint64_t dict_size, int prep_huffman, int64_t *d_quantized,
                              uint64_t *d_outlier_count, uint64_t *d_outlier_idx,
                              int64_t *d_outlier_val, uint64_t outlier_capacity, void *stream) {
  if (!h || !d_data || !d_quantized || !d_norm) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (!fusedc_ok(h) || h->force_v1)
    return fail(MGH_ERR_UNSUPPORTED_DIMENSION, "device-norm entry point needs the fused 3-D / 4-D path");
  if (prep_huffman && (!d_outlier_count || (outlier_capacity && (!d_outlier_idx || !d_outlier_val))))
    return fail(MGH_ERR_INVALID_ARGUMENT, "outlier buffers required with prep_huffman");
  HIP_TRY(hipSetDevice(h->device));
  if (ld_wanted(h, d_data, nullptr))
    return ld_entry(h, d_data, nullptr, true, stream, [&](const void *i, void *) {
      return mgh_decompose_quantize_dn(h, i, error_bound_type, tol, s, d_norm, num_subdomains, dict_size,
                                       prep_huffman, d_quantized, d_outlier_count, d_outlier_idx, d_outlier_val,
                                       outlier_capacity, stream);
    });
  return DISPATCH(h,
                  fused_q_entry_device<float>(h, (const float *)d_data, error_bound_type, tol, s,
                                              (const float *)d_norm, 1, num_subdomains, nullptr,
                                              dict_size, prep_huffman, d_quantized, d_outlier_count,
                                              d_outlier_idx, d_outlier_val, outlier_capacity,
                                              (hipStream_t)stream),
                  fused_q_entry_device<double>(h, (const double *)d_data, error_bound_type, tol, s,
                                               (const double *)d_norm, 1, num_subdomains, nullptr,
                                               dict_size, prep_huffman, d_quantized, d_outlier_count,
                                               d_outlier_idx, d_outlier_val, outlier_capacity,
                                               (hipStream_t)stream));
}

int mgh_dequantize_recompose(mgh_hierarchy *h, int64_t *d_quantized, int ebtype, double tol,
                             double s, double norm, uint64_t dict_size, int prep_huffman,
                             const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                             uint64_t outlier_count, void *d_data, void *stream) {
  if (!h || !d_data || !d_quantized) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  if (ld_wanted(h, nullptr, d_data))
    return ld_entry(h, nullptr, d_data, fusedc_ok(h) && !h->force_v1, stream, [&](const void *, void *o) {
      return mgh_dequantize_recompose(h, d_quantized, ebtype, tol, s, norm, dict_size, prep_huffman, d_outlier_idx,
                                      d_outlier_val, outlier_count, o, stream);
    });
  if (fusedc_ok(h) && !h->force_v1) {
    HIP_TRY(hipSetDevice(h->device));
    return DISPATCH(h,
                    dequantize_recompose_fused<float>(h, d_quantized, ebtype, tol, s, norm, dict_size,
                                                      prep_huffman, d_outlier_idx, d_outlier_val,
                                                      outlier_count, (float *)d_data,
                                                      (hipStream_t)stream),
                    dequantize_recompose_fused<double>(h, d_quantized, ebtype, tol, s, norm,
                                                       dict_size, prep_huffman, d_outlier_idx,
                                                       d_outlier_val, outlier_count,
                                                       (double *)d_data, (hipStream_t)stream));
  }
  int rc = mgh_dequantize(h, d_quantized, ebtype, tol, s, norm, dict_size, prep_huffman,
                          d_outlier_idx, d_outlier_val, outlier_count, d_data, stream);
  if (rc != MGH_SUCCESS) return rc;
  return mgh_recompose(h, d_data, d_data, stream);
}

#ifdef MGH_PHASE_TIMING
int mgh_debug_tail_read(unsigned long long *out64, int reset) {
  HIP_TRY(hipMemcpyFromSymbol(out64, HIP_SYMBOL(mgh::g_tail), sizeof(unsigned long long) * 64));
  if (reset) {
    unsigned long long z[64] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mgh::g_tail), z, sizeof z));
  }
  return MGH_SUCCESS;
}
int mgh_debug_phase_read(unsigned long long *out16, int reset) {
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(mgh::g_phase), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mgh::g_phase), z, sizeof z));
  }
  return MGH_SUCCESS;
}
#endif

int mgh_profile_enable(mgh_hierarchy *h, int enable) {
  if (!h) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  h->profiling = enable != 0;
  return MGH_SUCCESS;
}

int mgh_profile_filter(mgh_hierarchy *h, const char *name) {
  if (!h) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  h->prof_filter = name ? name : "";
  return MGH_SUCCESS;
}

int mgh_profile_read(mgh_hierarchy *h, const char **names, double *total_ms, uint64_t *launches,
                     int cap, int reset) {
  if (!h) return fail(MGH_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  int n = 0;
  for (auto &kv : h->prof) {
    ProfileEntry &e = kv.second;
    for (auto &ev : e.pending) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
      e.total_ms += ms;
      e.launches++;
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    e.pending.clear();
    if (n < cap) {
      if (names) names[n] = kv.first.c_str();
      if (total_ms) total_ms[n] = e.total_ms;
      if (launches) launches[n] = e.launches;
    }
    n++;
    if (reset) {
      e.total_ms = 0;
      e.launches = 0;
    }
  }
  return n;
}

int mgh_outlier_restore(int64_t *d_q, uint64_t n, const uint64_t *d_outlier_idx,
                        const int64_t *d_outlier_val, uint64_t outlier_count, void *stream) {
  if (!d_q || (outlier_count && (!d_outlier_idx || !d_outlier_val)))
    return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_outlier_restore: null argument");
  if (!outlier_count) return MGH_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  mgh::k_outlier_restore<<<(unsigned)((outlier_count + 255) / 256), 256, 0, st>>>(d_q, n, d_outlier_idx,
                                                                                 d_outlier_val, outlier_count);
  HIP_TRY(hipGetLastError());
  return MGH_SUCCESS;
}

int mgh_level_linearize(mgh_hierarchy *h, const int64_t *d_in, int64_t *d_out, int inverse,
                        uint64_t *d_outlier_idx, const uint64_t *d_outlier_count, uint64_t outlier_count,
                        uint64_t outlier_capacity, void *stream) {
  if (!h || !d_in || !d_out || d_in == d_out) return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_level_linearize: null or aliasing argument");
  if (inverse && d_outlier_idx) return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_level_linearize: indices are mapped forward only");
  if (h->L + 1 > mgh::kLinMaxLevels + 1) return fail(MGH_ERR_INVALID_ARGUMENT, "too many levels");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  mgh::LinMeta m{};
  const int *marks = nullptr;
  auto fill = [&](auto *hh, auto *ds) {
    m.D = hh->D;
    m.L = hh->L;
    for (int d = 0; d < hh->D; d++) {
      m.shape[d] = (uint32_t)hh->shape[d];
      m.markoff[d] = ds->qmeta.markoff[d];
      for (int l = 0; l <= hh->L; l++) m.lshape[l][d] = (uint32_t)hh->level_shape[l][d];
    }
    marks = ds->marks;
  };
  if (h->dtype == MGH_FLOAT) fill(HH<float>(h), DS<float>(h));
  else fill(HH<double>(h), DS<double>(h));
  const size_t total = h->total;
  const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
  TRY(launch(h, "level_linearize", st, [&] {
    if (inverse) mgh::k_level_linearize<int64_t, true><<<grid, 256, 0, st>>>(m, marks, total, d_in, d_out);
    else mgh::k_level_linearize<int64_t, false><<<grid, 256, 0, st>>>(m, marks, total, d_in, d_out);
  }));
  if (d_outlier_idx && (d_outlier_count || outlier_count)) {
    const unsigned long long cap = outlier_capacity ? outlier_capacity : (d_outlier_count ? ~0ull : outlier_count);
    const size_t work = d_outlier_count ? (size_t)std::min<unsigned long long>(cap, total) : (size_t)outlier_count;
    const unsigned g2 = (unsigned)std::max<size_t>(1, std::min<size_t>((work + 255) / 256, 4096));
    TRY(launch(h, "linearize_indices", st, [&] {
      mgh::k_linearize_indices<<<g2, 256, 0, st>>>(m, marks, total, d_outlier_idx,
                                                   (const unsigned long long *)d_outlier_count,
                                                   (unsigned long long)outlier_count, cap);
    }));
  }
  return MGH_SUCCESS;
}

int mgh_stream_calibrate(int dtype, const void *d_in, int64_t *d_out, void *d_side, uint64_t n,
                         int reps, double *ms_out, void *stream) {
  if (!d_in || !d_out || !d_side || !ms_out || n < 8 || reps < 1 || (dtype != MGH_FLOAT && dtype != MGH_DOUBLE)) {
    return fail(MGH_ERR_INVALID_ARGUMENT, "mgh_stream_calibrate: bad argument");
  }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  auto go = [&] {
    if (dtype == MGH_FLOAT)
      mgh::k_stream_mix<float><<<32768, 256, 0, st>>>((const float *)d_in, d_out, (float *)d_side,
                                                      (float *)d_side + n / 8, (size_t)n);
    else
      mgh::k_stream_mix<double><<<32768, 256, 0, st>>>((const double *)d_in, d_out, (double *)d_side,
                                                       (double *)d_side + n / 8, (size_t)n);
  };
  for (int i = 0; i < 2; i++) go();
  HIP_TRY(hipEventRecord(a, st));
  for (int i = 0; i < reps; i++) go();
  HIP_TRY(hipEventRecord(b, st));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  HIP_TRY(hipEventDestroy(a));
  HIP_TRY(hipEventDestroy(b));
  HIP_TRY(hipGetLastError());
  *ms_out = (double)ms / reps;
  return MGH_SUCCESS;
}

} // extern "C"
