// Tail kernel: all the small levels of the hierarchy in ONE launch.
//
// Below ~33^3 a level is pure launch latency (4-5 us per kernel, 4+ kernels per
// level, 5+ levels). Here a single 1024-thread workgroup walks those levels:
// per level the phases of kernels_v1.hpp (coefficients [+quantize], the three
// mass/restriction sweeps, the three Thomas solves, the correction) run back to
// back, separated by workgroup barriers only. Intermediate arrays are tiny and
// stay in L2; the load vector lives in LDS for the Thomas solves. Arithmetic is
// the shared element code of kernels_v1.hpp / kernels_ipk.hpp: bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"
#include "kernels_ipk.hpp"
#include "kernels_recompose.hpp"
#include "kernels_v1.hpp"

namespace mgh {

constexpr int kTailMaxLevels = 10;

// Developer build -DMGH_PHASE_TIMING: time (100 MHz wall clock ticks) thread 0 of k_tail spends in
// every phase, summed over the launches; read back with mgh_debug_tail_read (capi.hip).
// Slots: 0 tables + first level in, 1 solves of the level above; per tail level li, base 2 + 8 li:
// coefficients, f-, c-, r-sweep, f-, c-, r-solve, AddND; last slot 63: head.
#ifdef MGH_PHASE_TIMING
__device__ unsigned long long g_tail[64];
#define MGH_TT_DECL unsigned long long tt_t = wall_clock64();
#define MGH_TT(k) do { if (threadIdx.x == 0) { const unsigned long long tt_n = wall_clock64(); atomicAdd(&g_tail[(k) < 64 ? (k) : 63], tt_n - tt_t); tt_t = tt_n; } } while (0)
#else
#define MGH_TT_DECL
#define MGH_TT(k)
#endif

template <typename T> struct TailLevel {
  Box3 b;
  const T *ratio[3];
  const T *mass[3];
  const T *thomas[3];
  T quantizer, volume;
  int level;
};

template <typename T> struct TailArgs {
  int nlevels;  // lv[0] is the finest level handled here
  TailLevel<T> lv[kTailMaxLevels];
  const T *fine;  // nodal array of lv[0], strides (fI, fJ, 1)
  size_t fI, fJ;
  // The three Thomas solves + AddND of the level ABOVE lv[0] (whose coarse box is lv[0]'s fine
  // box), or nullptr: `fine` then holds that level's coarse nodes WITHOUT the correction,
  // pre_load its load vector (compact), pre_thomas the tables of its coarse grid -- two launches
  // and one round trip through global memory less (the box fits in LDS by construction).
  const T *pre_load;
  const T *pre_thomas[3];
  // Every table the kernel reads (ratios, mass constants, Thomas coefficients of its levels and
  // pre_thomas) lies in ONE contiguous block of the hierarchy's table arena: tab_base[0, tab_count).
  // It is copied to LDS in one pass, and a table at global address p is read at lds + (p - tab_base).
  const T *tab_base;
  uint32_t tab_count;
  T head_quantizer, head_volume;
  FusedArgs<T> out;    // coefficient / quantized output + outlier list
  // host-visible word (or nullptr) that receives the call's final outlier count: the host reads it
  // before a LATER call to pick the level kernel's variant (capi.hip: outlier_agg) -- a hint, read
  // without synchronisation
  unsigned long long *outliers_seen;
};

// LDS elements needed for a first tail level with fine box n and coarse box m
inline size_t tail_lds_elems(const Box3 &b) {
  const size_t nf = (size_t)b.n[0] * b.n[1] * b.n[2];
  const size_t mc = (size_t)b.m[0] * b.m[1] * b.m[2];
  return 2 * nf + (size_t)b.n[0] * b.n[1] * b.m[2] + (size_t)b.n[0] * b.m[1] * b.m[2] + 2 * mc;
}

constexpr int kTailQpMax = 40;  // = kMaxLevels (kernels_fused.hpp)
// bytes of the LDS header of k_tail: table offsets + quantizer table
template <typename T> constexpr size_t tail_header_bytes() {
  return (kTailMaxLevels * 9 * sizeof(uint32_t) + 2 * kTailQpMax * sizeof(T) + 15) / 16 * 16;
}
// LDS elements of the tables (ratios, mass constants, Thomas coefficients) of one tail level
inline size_t tail_table_elems(const Box3 &b) {
  return (size_t)b.n[0] + b.n[1] + b.n[2] + 12 * ((size_t)b.m[0] + b.m[1] + b.m[2]);
}

template <typename T, int OUT>
__global__ void __launch_bounds__(1024)
k_tail(TailArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t tid = threadIdx.x, NT = blockDim.x;
  FusedArgs<T> O = A.out;
  MGH_TT_DECL
  // Every phase below reads its tables per element; out of global memory each phase would start
  // with a dependent round trip of a microsecond or two (12 us per level measured). All tables of
  // the tail levels are a few KB: they go to LDS once, behind the data regions.
  // (header of the dynamic LDS: a static __shared__ array would count against the 160 KB)
  T *qps = reinterpret_cast<T *>(smem_raw + kTailMaxLevels * 9 * sizeof(uint32_t));
  T *tab;
  // carve LDS for the first (largest) level; deeper levels reuse the same regions
  T *X, *Y, *C, *T1, *T2, *T3;
  {
    const Box3 &b = A.lv[0].b;
    const size_t nf = (size_t)b.n[0] * b.n[1] * b.n[2];
    const size_t mc = (size_t)b.m[0] * b.m[1] * b.m[2];
    T *base = reinterpret_cast<T *>(smem_raw + tail_header_bytes<T>());
    X = base;
    C = X + nf;
    T1 = C + nf;
    T2 = T1 + (size_t)b.n[0] * b.n[1] * b.m[2];
    T3 = T2 + (size_t)b.n[0] * b.m[1] * b.m[2];
    Y = T3 + mc;
    tab = Y + mc;
    // One round of loads for everything the kernel starts from -- quantizer table, the table block
    // (phase-wise table loads from global memory cost a dependent round trip each: 36 small loops
    // took 12 of the kernel's 62 us), the nodal values of the first level and the load vector of
    // the level above -- all in flight together, then one barrier.
    if (O.qp)
      for (uint32_t e = tid; e < 2u * (uint32_t)O.nlev && e < 2u * kMaxLevels; e += NT) qps[e] = O.qp[e];
    for (uint32_t e = tid; e < A.tab_count; e += NT) tab[e] = A.tab_base[e];
    const uint32_t n0 = b.n[0], n1 = b.n[1], n2 = b.n[2];
    for (uint32_t e = tid; e < nf; e += NT) {
      const uint32_t k = e % n2, j = (e / n2) % n1, i = e / (n2 * n1);
      X[e] = A.fine[i * A.fI + j * A.fJ + k];
    }
    if (A.pre_load)
      for (uint32_t e = tid; e < nf; e += NT) C[e] = A.pre_load[e];
    __syncthreads();
    MGH_TT(0);
    if (A.pre_load) {
      // correction of the level above: f-, c-, r-solve of its load vector (IPKFunctor.h:127,147),
      // added to the coarse nodes (LevelwiseProcessingKernel.hpp:69-74)
      const T *pt0 = tab + (A.pre_thomas[0] - A.tab_base), *pt1 = tab + (A.pre_thomas[1] - A.tab_base),
              *pt2 = tab + (A.pre_thomas[2] - A.tab_base);
      for (uint32_t p = tid; p < n0 * n1; p += NT) thomas_lds<T, false>(C + (size_t)p * n2, 1, n2, pt2);
      __syncthreads();
      for (uint32_t p = tid; p < n0 * n2; p += NT)
        thomas_lds<T, false>(C + (size_t)(p / n2) * n1 * n2 + (p % n2), n2, n1, pt1);
      __syncthreads();
      for (uint32_t p = tid; p < n1 * n2; p += NT) thomas_lds<T, false>(C + p, n1 * n2, n0, pt0);
      __syncthreads();
      for (uint32_t e = tid; e < nf; e += NT) X[e] = X[e] + C[e];
    }
  }
  __syncthreads();
  MGH_TT(1);
  for (int li = 0; li < A.nlevels; li++) {
    const TailLevel<T> &L = A.lv[li];
    const Box3 b = L.b;
    const uint32_t n0 = b.n[0], n1 = b.n[1], n2 = b.n[2];
    const uint32_t m0 = b.m[0], m1 = b.m[1], m2 = b.m[2];
    const bool qp_lds = O.qp && O.nlev <= kMaxLevels;
    O.quantizer = qp_lds ? qps[L.level] : (O.qp ? O.qp[L.level] : L.quantizer);
    O.volume = qp_lds ? qps[O.nlev + L.level] : (O.qp ? O.qp[O.nlev + L.level] : L.volume);
    const T *ratio0 = tab + (L.ratio[0] - A.tab_base), *ratio1 = tab + (L.ratio[1] - A.tab_base),
            *ratio2 = tab + (L.ratio[2] - A.tab_base);
    const T *mass0 = tab + (L.mass[0] - A.tab_base), *mass1 = tab + (L.mass[1] - A.tab_base),
            *mass2 = tab + (L.mass[2] - A.tab_base);
    const T *thom0 = tab + (L.thomas[0] - A.tab_base), *thom1 = tab + (L.thomas[1] - A.tab_base),
            *thom2 = tab + (L.thomas[2] - A.tab_base);
    // ---- coefficients (+ output) and coarse nodes ----
    {
      const uint32_t total = n0 * n1 * n2;
      for (uint32_t e0 = 0; e0 < total; e0 += NT) {
        const uint32_t e = e0 + tid;
        const bool live = e < total;
        const uint32_t ee = live ? e : 0;
        const uint32_t k = ee % n2, j = (ee / n2) % n1, i = ee / (n2 * n1);
        bool is_coarse;
        const T v = gpk_reo_elem(b, (const T *)X, (size_t)n1 * n2, (size_t)n2, ratio0, ratio1,
                                 ratio2, i, j, k, is_coarse);
        const size_t lin = (size_t)i * O.dI + (size_t)j * O.dJ + k;
        if (live) {
          if (is_coarse)
            Y[((size_t)i * m1 + j) * m2 + k] = v;
          else
            C[ee] = v;
        }
        const bool emit = live && !is_coarse;
        if (OUT == OUT_T) {
          if (emit) O.coef[lin] = v;
        } else {
          const T vv[1] = {v};
          const size_t ll[1] = {lin};
          const bool on[1] = {emit};
          emit_quantized<T, 1>(O, vv, ll, on);
        }
      }
    }
    __syncthreads();
    MGH_TT(2 + 8 * li + 0);
    // ---- mass/restriction sweeps: f, c, r ----
    {
      const uint32_t total = n0 * n1 * m2;
      for (uint32_t e = tid; e < total; e += NT) {
        const uint32_t k = e % m2, j = (e / m2) % n1, i = e / (m2 * n1);
        T1[e] = lpk_elem<T, 2>(n2, m2, (const T *)C, (size_t)n1 * n2, (size_t)n2, mass2, m0, m1, i,
                               j, k);
      }
    }
    __syncthreads();
    MGH_TT(2 + 8 * li + 1);
    {
      const uint32_t total = n0 * m1 * m2;
      for (uint32_t e = tid; e < total; e += NT) {
        const uint32_t k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
        T2[e] = lpk_elem<T, 1>(n1, m1, (const T *)T1, (size_t)n1 * m2, (size_t)m2, mass1, 0, 0, i, j,
                               k);
      }
    }
    __syncthreads();
    MGH_TT(2 + 8 * li + 2);
    const uint32_t mtot = m0 * m1 * m2;
    for (uint32_t e = tid; e < mtot; e += NT) {
      const uint32_t k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
      T3[e] = lpk_elem<T, 0>(n0, m0, (const T *)T2, (size_t)m1 * m2, (size_t)m2, mass0, 0, 0, i, j, k);
    }
    __syncthreads();
    MGH_TT(2 + 8 * li + 3);
    // ---- Thomas solves in LDS: f, c, r ----
    for (uint32_t p = tid; p < m0 * m1; p += NT) thomas_lds<T, false>(T3 + (size_t)p * m2, 1, m2, thom2);
    __syncthreads();
    MGH_TT(2 + 8 * li + 4);
    for (uint32_t p = tid; p < m0 * m2; p += NT)
      thomas_lds<T, false>(T3 + (size_t)(p / m2) * m1 * m2 + (p % m2), m2, m1, thom1);
    __syncthreads();
    MGH_TT(2 + 8 * li + 5);
    for (uint32_t p = tid; p < m1 * m2; p += NT) thomas_lds<T, false>(T3 + p, m1 * m2, m0, thom0);
    __syncthreads();
    MGH_TT(2 + 8 * li + 6);
    // ---- apply the correction (AddND); the corrected coarse nodes are the next level ----
    for (uint32_t e = tid; e < mtot; e += NT) Y[e] += T3[e];
    __syncthreads();
    MGH_TT(2 + 8 * li + 7);
    T *tmp = X;
    X = Y;
    Y = tmp;
  }
  // ---- head: level-0 nodal values (now in X) ----
  {
    const TailLevel<T> &L = A.lv[A.nlevels - 1];
    const uint32_t m0 = L.b.m[0], m1 = L.b.m[1], m2 = L.b.m[2];
    const uint32_t total = m0 * m1 * m2;
    const bool qp_lds = O.qp && O.nlev <= kMaxLevels;
    O.quantizer = qp_lds ? qps[0] : (O.qp ? O.qp[0] : A.head_quantizer);
    O.volume = qp_lds ? qps[O.nlev] : (O.qp ? O.qp[O.nlev] : A.head_volume);
    for (uint32_t e0 = 0; e0 < total; e0 += NT) {
      const uint32_t e = e0 + tid;
      const bool live = e < total;
      const uint32_t ee = live ? e : 0;
      const uint32_t k = ee % m2, j = (ee / m2) % m1, i = ee / (m2 * m1);
      const size_t lin = (size_t)i * O.dI + (size_t)j * O.dJ + k;
      const T v = X[ee];
      if (OUT == OUT_T) {
        if (live) O.coef[lin] = v;
      } else {
        const T vv[1] = {v};
        const size_t ll[1] = {lin};
        const bool on[1] = {live};
        emit_quantized<T, 1>(O, vv, ll, on);
      }
    }
  }
  if (OUT != OUT_T && A.outliers_seen && O.outlier_count) {  // (this kernel is the last of the call)
    __syncthreads();
    if (tid == 0)
      *A.outliers_seen = __hip_atomic_load(O.outlier_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  MGH_TT(63);
}

// ---------------------------------------------------------------------------------------
// The mirror for decompression: the small levels of the recomposition (coarsest first) in ONE
// launch -- level-0 nodal values out of the head of the coefficient array, then per level the
// dequantized coefficient field, the three mass/restriction sweeps, the three Thomas solves, the
// subtraction of the correction from the coarse nodes and the node restore (kernels_v1.hpp
// element code; k_level_loadvec_q / k_level_restore_q compute the same values). Without it
// every one of these levels costs four dependent launches of a few microseconds each.
// ---------------------------------------------------------------------------------------
template <typename T> struct HeadLevel {
  Box3 b;
  const T *ratio[3];
  const T *mass[3];
  const T *thomas[3];
  T qv;  // dequantize factor of the level
};

template <typename T> struct HeadArgs {
  int nlevels;  // lv[0] is level 1 (the coarsest), lv[nlevels - 1] the finest handled here
  HeadLevel<T> lv[kTailMaxLevels];
  T qv0;        // dequantize factor of level 0
  RecomposeArgs<T> in;  // coefficient source (q / coef / q16, strides dI dJ, half, outlier table)
  T *out;       // nodal values of the finest level handled, strides (oI, oJ, 1)
  size_t oI, oJ;
  // the tables of all its levels: one contiguous block of the hierarchy's arena, copied to LDS in
  // one pass (as in k_tail); a table at global address p is read at lds + (p - tab_base)
  const T *tab_base;
  uint32_t tab_count;
};

inline size_t head_lds_elems(const Box3 &b) {
  const size_t nf = (size_t)b.n[0] * b.n[1] * b.n[2];
  const size_t mc = (size_t)b.m[0] * b.m[1] * b.m[2];
  return 3 * nf + (size_t)b.n[0] * b.n[1] * b.m[2] + (size_t)b.n[0] * b.m[1] * b.m[2] + mc;
}

template <typename T, typename QT>
__global__ void __launch_bounds__(1024)
k_recompose_head(HeadArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t tid = threadIdx.x, NT = blockDim.x;
  RecomposeArgs<T> Q = A.in;
  T *X, *Y, *C, *T1, *T2, *T3, *tab;
  {
    const Box3 &b = A.lv[A.nlevels - 1].b;  // the largest level: deeper ones reuse its regions
    const size_t nf = (size_t)b.n[0] * b.n[1] * b.n[2];
    T *base = reinterpret_cast<T *>(smem_raw);
    X = base;
    Y = X + nf;
    C = Y + nf;
    T1 = C + nf;
    T2 = T1 + (size_t)b.n[0] * b.n[1] * b.m[2];
    T3 = T2 + (size_t)b.n[0] * b.m[1] * b.m[2];
    tab = T3 + (size_t)b.m[0] * b.m[1] * b.m[2];
  }
  // (every phase below reads its tables per element: from LDS, not with a round trip to global
  // memory at the start of each of them)
  for (uint32_t e = tid; e < A.tab_count; e += NT) tab[e] = A.tab_base[e];
  {  // level-0 nodal values
    const Box3 &b = A.lv[0].b;
    const uint32_t m0 = b.m[0], m1 = b.m[1], m2 = b.m[2];
    Q.qv = A.qv0;
    for (uint32_t e = tid; e < m0 * m1 * m2; e += NT) {
      const uint32_t k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
      const size_t lin = (size_t)i * Q.dI + (size_t)j * Q.dJ + k;
      X[e] = qdecode(Q, qload<T>(Q, qsrc<T>(Q, QT()) + lin), lin);
    }
  }
  __syncthreads();
  for (int li = 0; li < A.nlevels; li++) {
    const HeadLevel<T> &L = A.lv[li];
    const Box3 b = L.b;
    const uint32_t n0 = b.n[0], n1 = b.n[1], n2 = b.n[2];
    const uint32_t m0 = b.m[0], m1 = b.m[1], m2 = b.m[2];
    const uint32_t nf = n0 * n1 * n2;
    Q.qv = L.qv;
    const T *mass0 = tab + (L.mass[0] - A.tab_base), *mass1 = tab + (L.mass[1] - A.tab_base),
            *mass2 = tab + (L.mass[2] - A.tab_base);
    const T *thom0 = tab + (L.thomas[0] - A.tab_base), *thom1 = tab + (L.thomas[1] - A.tab_base),
            *thom2 = tab + (L.thomas[2] - A.tab_base);
    const T *ratio0 = tab + (L.ratio[0] - A.tab_base), *ratio1 = tab + (L.ratio[1] - A.tab_base),
            *ratio2 = tab + (L.ratio[2] - A.tab_base);
    // ---- dequantized coefficient field in the reordered layout of the level's box (the coarse
    // corner is never read as a coefficient: the f-sweep takes it as zero, the restore skips it)
    for (uint32_t e = tid; e < nf; e += NT) {
      const uint32_t k = e % n2, j = (e / n2) % n1, i = e / (n2 * n1);
      const bool corner = i < m0 && j < m1 && k < m2;
      const size_t lin = (size_t)i * Q.dI + (size_t)j * Q.dJ + k;
      C[e] = corner ? (T)0 : qdecode(Q, qload<T>(Q, qsrc<T>(Q, QT()) + lin), lin);
    }
    __syncthreads();
    // ---- mass/restriction sweeps: f, c, r ----
    for (uint32_t e = tid; e < n0 * n1 * m2; e += NT) {
      const uint32_t k = e % m2, j = (e / m2) % n1, i = e / (m2 * n1);
      T1[e] = lpk_elem<T, 2>(n2, m2, (const T *)C, (size_t)n1 * n2, (size_t)n2, mass2, m0, m1, i, j, k);
    }
    __syncthreads();
    for (uint32_t e = tid; e < n0 * m1 * m2; e += NT) {
      const uint32_t k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
      T2[e] = lpk_elem<T, 1>(n1, m1, (const T *)T1, (size_t)n1 * m2, (size_t)m2, mass1, 0, 0, i, j, k);
    }
    __syncthreads();
    const uint32_t mtot = m0 * m1 * m2;
    for (uint32_t e = tid; e < mtot; e += NT) {
      const uint32_t k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
      T3[e] = lpk_elem<T, 0>(n0, m0, (const T *)T2, (size_t)m1 * m2, (size_t)m2, mass0, 0, 0, i, j, k);
    }
    __syncthreads();
    // ---- Thomas solves: f, c, r ----
    for (uint32_t p = tid; p < m0 * m1; p += NT) thomas_lds<T, false>(T3 + (size_t)p * m2, 1, m2, thom2);
    __syncthreads();
    for (uint32_t p = tid; p < m0 * m2; p += NT)
      thomas_lds<T, false>(T3 + (size_t)(p / m2) * m1 * m2 + (p % m2), m2, m1, thom1);
    __syncthreads();
    for (uint32_t p = tid; p < m1 * m2; p += NT) thomas_lds<T, false>(T3 + p, m1 * m2, m0, thom0);
    __syncthreads();
    // ---- subtract the correction from the coarse nodes (SubtractND) ----
    for (uint32_t e = tid; e < mtot; e += NT) X[e] = X[e] - T3[e];
    __syncthreads();
    // ---- node restore: the fine nodal values of this level = the coarse nodes of the next ----
    for (uint32_t e = tid; e < nf; e += NT) {
      const uint32_t k = e % n2, j = (e / n2) % n1, i = e / (n2 * n1);
      uint32_t rp, cp, fp;
      const T v = gpk_rev_elem(b, (const T *)X, (const T *)C, (size_t)n1 * n2, (size_t)n2, ratio0,
                               ratio1, ratio2, i, j, k, rp, cp, fp);
      Y[((size_t)rp * n1 + cp) * n2 + fp] = v;
    }
    __syncthreads();
    T *tmp = X;
    X = Y;
    Y = tmp;
  }
  {
    const Box3 &b = A.lv[A.nlevels - 1].b;
    const uint32_t n0 = b.n[0], n1 = b.n[1], n2 = b.n[2];
    for (uint32_t e = tid; e < n0 * n1 * n2; e += NT) {
      const uint32_t k = e % n2, j = (e / n2) % n1, i = e / (n2 * n1);
      A.out[i * A.oI + j * A.oJ + k] = X[e];
    }
  }
}

} // namespace mgh
