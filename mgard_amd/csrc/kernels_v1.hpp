// Straightforward (one thread per output element) HIP kernels for the MGARD-X
// decomposition chain. They are the correctness baseline of this library: every
// arithmetic expression keeps the reference's operation order, and the file is
// compiled with -ffp-contract=off so no FMA is formed (the reference never
// defines MGARD_X_FMA). Tuned kernels (kernels_fast.hpp) must match these
// bit-for-bit.
//
// Layout vocabulary: a level-l "nodal" array is the natural-order box
// (nr, nc, nf); the "reordered" layout puts the coarse nodes first along every
// dim ([0,rr) x [0,cc) x [0,ff)) and the coefficients behind them, exactly like
// the sub-array views of
// include/mgard-x/DataRefactoring/MultiDimension/Coefficient/CalcCoefficients3D.hpp:51-73.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgh {

// Coefficient/GPKFunctor.h:21-23 (non-FMA branch)
template <typename T> __device__ __forceinline__ T lerp_ref(T v0, T v1, T t) {
  T r = v0 + v0 * t * (T)-1;
  r = r + t * v1;
  return r;
}

struct Box3 {
  uint32_t n[3];  // fine sizes (r, c, f)
  uint32_t m[3];  // coarse sizes (rr, cc, ff)
};

// fine index of reordered index i along a dim: coarse node -> min(2i, n-1),
// coefficient -> odd node 2(i-m)+1
__device__ __forceinline__ uint32_t fine_pos(uint32_t i, uint32_t n, uint32_t m, bool &odd) {
  odd = i >= m;
  if (odd) return 2 * (i - m) + 1;
  uint32_t p = 2 * i;
  return p < n - 1 ? p : n - 1;
}

// ---------------------------------------------------------------------------
// GPK: coefficient computation + reordering. Restates GpkReo3D
// (Coefficient/GridProcessingKernel3D.hpp:21-1179). src: natural fine box with
// element strides (sI, sJ, 1). Coarse nodes go to `coarse` (compact m[0..2]),
// coefficients to `dst` (strides dI, dJ, 1) at their reordered position.
// When dst_coarse_too != 0 the coarse nodes are ALSO written to dst (used on
// the last level so that dst holds the complete result).
// ---------------------------------------------------------------------------
// value of reordered node (i, j, k): the raw node for an all-coarse position
// (is_coarse = true), else node minus the interpolant of its coarse neighbours
template <typename T>
__device__ __forceinline__ T gpk_reo_elem(const Box3 &b, const T *__restrict__ src, size_t sI,
                                          size_t sJ, const T *__restrict__ ratio_r,
                                          const T *__restrict__ ratio_c,
                                          const T *__restrict__ ratio_f, uint32_t i, uint32_t j,
                                          uint32_t k, bool &is_coarse) {
  bool ro, co, fo;
  const uint32_t rp = fine_pos(i, b.n[0], b.m[0], ro);
  const uint32_t cp = fine_pos(j, b.n[1], b.m[1], co);
  const uint32_t fp = fine_pos(k, b.n[2], b.m[2], fo);
  const T center = src[rp * sI + cp * sJ + fp];
  is_coarse = !ro && !co && !fo;
  if (is_coarse) return center;
  const uint32_t r0 = ro ? rp - 1 : rp, r1 = rp + 1;
  const uint32_t c0 = co ? cp - 1 : cp, c1 = cp + 1;
  const uint32_t f0 = fo ? fp - 1 : fp, f1 = fp + 1;
  T hr[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    if (a == 1 && !ro) break;
    const uint32_t ri = a ? r1 : r0;
    T gc[2];
#pragma unroll
    for (int bb = 0; bb < 2; bb++) {
      if (bb == 1 && !co) break;
      const uint32_t ci = bb ? c1 : c0;
      const T *row = src + ri * sI + ci * sJ;
      gc[bb] = fo ? lerp_ref(row[f0], row[f1], ratio_f[f0]) : row[f0];
    }
    hr[a] = co ? lerp_ref(gc[0], gc[1], ratio_c[c0]) : gc[0];
  }
  const T res = ro ? lerp_ref(hr[0], hr[1], ratio_r[r0]) : hr[0];
  return center - res;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_gpk_reo(Box3 b, const T *__restrict__ src, size_t sI, size_t sJ, T *__restrict__ coarse,
          T *__restrict__ dst, size_t dI, size_t dJ, const T *__restrict__ ratio_r,
          const T *__restrict__ ratio_c, const T *__restrict__ ratio_f) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
  const uint32_t i = blockIdx.z;
  if (k >= b.n[2] || j >= b.n[1] || i >= b.n[0]) return;
  bool is_coarse;
  const T v = gpk_reo_elem(b, src, sI, sJ, ratio_r, ratio_c, ratio_f, i, j, k, is_coarse);
  if (is_coarse)
    coarse[((size_t)i * b.m[1] + j) * b.m[2] + k] = v;
  else
    dst[i * dI + j * dJ + k] = v;
}

// GpkRev3D (GridProcessingKernel3D.hpp:1231-2352): natural fine box out of
// (coarse compact, coefficients in `coef` reordered layout). One thread per
// fine node; odd nodes add the interpolant of their coarse neighbours.
// value of the fine node that reordered position (i, j, k) stands for, and its natural position
template <typename T>
__device__ __forceinline__ T gpk_rev_elem(const Box3 &b, const T *__restrict__ coarse,
                                          const T *__restrict__ coef, size_t cI, size_t cJ,
                                          const T *__restrict__ ratio_r, const T *__restrict__ ratio_c,
                                          const T *__restrict__ ratio_f, uint32_t i, uint32_t j,
                                          uint32_t k, uint32_t &rp, uint32_t &cp, uint32_t &fp) {
  bool ro, co, fo;
  rp = fine_pos(i, b.n[0], b.m[0], ro);
  cp = fine_pos(j, b.n[1], b.m[1], co);
  fp = fine_pos(k, b.n[2], b.m[2], fo);
  const size_t mJ = b.m[2], mI = (size_t)b.m[1] * b.m[2];
  // coarse index of the even fine position p (or of the padded last node)
  auto cidx = [](uint32_t p, uint32_t n, uint32_t m) -> uint32_t {
    return (p == n - 1) ? m - 1 : p / 2;
  };
  if (!ro && !co && !fo) return coarse[i * mI + j * mJ + k];
  const uint32_t r0 = cidx(ro ? rp - 1 : rp, b.n[0], b.m[0]), r1 = cidx(rp + 1, b.n[0], b.m[0]);
  const uint32_t c0 = cidx(co ? cp - 1 : cp, b.n[1], b.m[1]), c1 = cidx(cp + 1, b.n[1], b.m[1]);
  const uint32_t f0 = cidx(fo ? fp - 1 : fp, b.n[2], b.m[2]), f1 = cidx(fp + 1, b.n[2], b.m[2]);
  T hr[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    if (a == 1 && !ro) break;
    const uint32_t ri = a ? r1 : r0;
    T gc[2];
#pragma unroll
    for (int bb = 0; bb < 2; bb++) {
      if (bb == 1 && !co) break;
      const uint32_t ci = bb ? c1 : c0;
      const T *row = coarse + ri * mI + ci * mJ;
      gc[bb] = fo ? lerp_ref(row[f0], row[f1], ratio_f[fp - 1]) : row[f0];
    }
    hr[a] = co ? lerp_ref(gc[0], gc[1], ratio_c[cp - 1]) : gc[0];
  }
  T res = coef[i * cI + j * cJ + k];
  res += ro ? lerp_ref(hr[0], hr[1], ratio_r[rp - 1]) : hr[0];
  return res;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_gpk_rev(Box3 b, const T *__restrict__ coarse, const T *__restrict__ coef, size_t cI, size_t cJ,
          T *__restrict__ out, size_t oI, size_t oJ, const T *__restrict__ ratio_r,
          const T *__restrict__ ratio_c, const T *__restrict__ ratio_f) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
  const uint32_t i = blockIdx.z;
  if (k >= b.n[2] || j >= b.n[1] || i >= b.n[0]) return;
  uint32_t rp, cp, fp;
  const T res = gpk_rev_elem(b, coarse, coef, cI, cJ, ratio_r, ratio_c, ratio_f, i, j, k, rp, cp, fp);
  out[rp * oI + cp * oJ + fp] = res;
}

// ---------------------------------------------------------------------------
// LPK: fused mass matrix * restriction along one axis. Restates
// Lpk1/2/3Reo3D (Correction/LinearProcessingKernel3D.hpp:27-400, 449-717,
// 762-1048) with mass_trans (Correction/LPKFunctor.h:77-93) on host-prepared
// per-node constants (hierarchy.hpp: mass_table).
// in: (n0,n1,n2) box in reordered layout along AXIS (even part [0,m), odd part
// [m, n)), strides (iI, iJ, 1). out: same box with AXIS shrunk to m, strides
// (oI, oJ, 1). zero_i/zero_j (LPK1 only): rows with i < zero_i && j < zero_j
// read their even part as 0 (zero_r/zero_c/zero_f, :98-101).
// ---------------------------------------------------------------------------
template <typename T, int AXIS>
__device__ __forceinline__ T lpk_elem(uint32_t n, uint32_t m, const T *__restrict__ in, size_t iI,
                                      size_t iJ, const T *__restrict__ mt, uint32_t zero_i,
                                      uint32_t zero_j, uint32_t i, uint32_t j, uint32_t k) {
  const uint32_t q = AXIS == 0 ? i : (AXIS == 1 ? j : k); // coarse index along AXIS
  const size_t st = AXIS == 0 ? iI : (AXIS == 1 ? iJ : 1);
  const size_t base = (AXIS == 0 ? 0 : i * iI) + (AXIS == 1 ? 0 : j * iJ) + (AXIS == 2 ? 0 : k);
  const uint32_t nodd = n - m;
  const bool ez = (AXIS == 2) && (i < zero_i) && (j < zero_j);
  const T *pe = in + base;            // even part
  const T *po = in + base + m * st;   // odd part
  const T a = (q >= 1 && !ez) ? pe[(q - 1) * st] : (T)0;
  const T bq = (q >= 1 && q - 1 < nodd) ? po[(q - 1) * st] : (T)0;
  const T c = ez ? (T)0 : pe[q * st];
  const T d = (q < nodd) ? po[q * st] : (T)0;
  const T e = (q + 1 < m && !ez) ? pe[(q + 1) * st] : (T)0;
  const T w0 = mt[0 * m + q], w1 = mt[1 * m + q], w2 = mt[2 * m + q], w3 = mt[3 * m + q],
          w4 = mt[4 * m + q], w5 = mt[5 * m + q], w6 = mt[6 * m + q], r1 = mt[7 * m + q],
          r4 = mt[8 * m + q];
  const T tb = a * w0 + bq * w1 + c * w2;
  T tc = bq * w2 + c * w3 + d * w4;
  const T td = c * w4 + d * w5 + e * w6;
  tc += tb * r1 + td * r4;
  return tc;
}

template <typename T, int AXIS>
__global__ void __launch_bounds__(256)
k_lpk(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n, uint32_t m, const T *__restrict__ in,
      size_t iI, size_t iJ, T *__restrict__ out, size_t oI, size_t oJ,
      const T *__restrict__ mt, uint32_t zero_i, uint32_t zero_j) {
  // output extents
  const uint32_t e0 = AXIS == 0 ? m : n0, e1 = AXIS == 1 ? m : n1, e2 = AXIS == 2 ? m : n2;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
  const uint32_t i = blockIdx.z;
  if (k >= e2 || j >= e1 || i >= e0) return;
  out[i * oI + j * oJ + k] = lpk_elem<T, AXIS>(n, m, in, iI, iJ, mt, zero_i, zero_j, i, j, k);
}

// ---------------------------------------------------------------------------
// IPK: Thomas solve along one axis, in place on a compact (n0,n1,n2) box.
// Restates Ipk1/2/3Reo3D (Correction/IterativeProcessingKernel3D.hpp:28-371,
// 418-745, 792-1169) with tridiag_forward2/backward2 (IPKFunctor.h:127,147).
// tt: thomas_table (hierarchy.hpp) of the solved dim: [0,n) forward
// multiplier, [n,2n) backward am, [2n,3n) backward bm.
// One thread per pencil. If add_to != nullptr the solution is ADDED (sign=+1)
// or SUBTRACTED (sign=-1) into add_to (compact, same shape) during the backward
// sweep instead of being stored (AddND / SubtractND,
// CopyND/LevelwiseProcessingKernel.hpp:69-74).
// ---------------------------------------------------------------------------
template <typename T, int AXIS>
__global__ void __launch_bounds__(256)
k_ipk(uint32_t n0, uint32_t n1, uint32_t n2, T *__restrict__ x, const T *__restrict__ tt,
      T *__restrict__ add_to, int sign) {
  const uint32_t n = AXIS == 0 ? n0 : (AXIS == 1 ? n1 : n2);
  // pencil coordinates: the two non-AXIS dims; fastest-varying thread index maps
  // to the fastest non-AXIS dim
  const uint32_t pa = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pb = blockIdx.y * blockDim.y + threadIdx.y;
  uint32_t ea, eb;
  size_t base, st;
  const size_t sI = (size_t)n1 * n2, sJ = n2;
  if (AXIS == 2) { // pencils over (i=pb, j=pa)
    ea = n1; eb = n0;
    if (pa >= ea || pb >= eb) return;
    base = pb * sI + pa * sJ; st = 1;
  } else if (AXIS == 1) { // pencils over (i=pb, k=pa)
    ea = n2; eb = n0;
    if (pa >= ea || pb >= eb) return;
    base = pb * sI + pa; st = sJ;
  } else { // pencils over (j=pb, k=pa)
    ea = n2; eb = n1;
    if (pa >= ea || pb >= eb) return;
    base = pb * sJ + pa; st = sI;
  }
  T *p = x + base;
  // The loads of a pencil do not depend on the recurrence: issue them U at a time ahead of the
  // dependent chain (a load per step would expose the full memory latency at every step --
  // this kernel serves the pencils that are too long for the LDS-staged solves).
  constexpr uint32_t U = 8;
  T prev = 0;
  uint32_t i = 0;
  for (; i + U <= n; i += U) {
    T v[U], m[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      v[u] = p[(size_t)(i + u) * st];
      m[u] = tt[i + u];
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      v[u] = v[u] - prev * m[u];
      prev = v[u];
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) p[(size_t)(i + u) * st] = v[u];
  }
  for (; i < n; i++) {
    T cur = p[(size_t)i * st];
    cur = cur - prev * tt[i];
    p[(size_t)i * st] = cur;
    prev = cur;
  }
  prev = 0;
  T *q = add_to ? add_to + base : nullptr;
  int64_t k = (int64_t)n - 1;
  for (; k >= (int64_t)U - 1; k -= U) {
    T v[U], am[U], bm[U], o[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      v[u] = p[(size_t)(k - u) * st];
      am[u] = tt[n + (k - u)];
      bm[u] = tt[2 * n + (k - u)];
      o[u] = q ? q[(size_t)(k - u) * st] : (T)0;
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      v[u] = (v[u] - am[u] * prev) / bm[u];
      prev = v[u];
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      if (q) q[(size_t)(k - u) * st] = sign > 0 ? o[u] + v[u] : o[u] - v[u];
      else p[(size_t)(k - u) * st] = v[u];
    }
  }
  for (; k >= 0; k--) {
    T cur = p[(size_t)k * st];
    cur = (cur - tt[n + k] * prev) / tt[2 * n + k];
    if (q) {
      T *qq = q + (size_t)k * st;
      if (sign > 0) *qq += cur; else *qq -= cur;
    } else {
      p[(size_t)k * st] = cur;
    }
    prev = cur;
  }
}

// dst (strides dI,dJ) <- src (strides sI,sJ) over an (n0,n1,n2) box
template <typename T>
__global__ void __launch_bounds__(256)
k_copy_box(uint32_t n0, uint32_t n1, uint32_t n2, const T *__restrict__ src, size_t sI, size_t sJ,
           T *__restrict__ dst, size_t dI, size_t dJ) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
  const uint32_t i = blockIdx.z;
  if (k >= n2 || j >= n1 || i >= n0) return;
  dst[i * dI + j * dJ + k] = src[i * sI + j * sJ + k];
}

// ---------------------------------------------------------------------------
// Level-wise linear quantizer (Quantization/LinearQuantization.hpp:146-264).
// marks: D arrays concatenated (offset markoff[d]) giving the level of an index
// along dim d; qz[level] = quantizer (reciprocal for quantize), vol[level] =
// sqrt(prod volumes) or 1. shape/strides describe the dense array, last dim
// fastest; idx decomposition from the linear index.
// ---------------------------------------------------------------------------
struct QuantMeta {
  int D;
  int calc_vol;
  uint32_t shape[5];
  uint32_t markoff[5];
};

template <typename T> __device__ __forceinline__ T abs_t(T x);
template <> __device__ __forceinline__ float abs_t<float>(float x) { return fabsf(x); }
template <> __device__ __forceinline__ double abs_t<double>(double x) { return fabs(x); }
template <typename T> __device__ __forceinline__ T copysign_t(T x, T y);
template <> __device__ __forceinline__ float copysign_t<float>(float x, float y) { return copysignf(x, y); }
template <> __device__ __forceinline__ double copysign_t<double>(double x, double y) { return copysign(x, y); }

template <typename T>
__device__ __forceinline__ int64_t quantize_one(T t, T quantizer, T volume) {
  return (int64_t)copysign_t((T)0.5 + abs_t(t * quantizer * volume), t);
}

__device__ __forceinline__ int level_of(const QuantMeta &m, const int *__restrict__ marks,
                                        size_t lin) {
  int level = 0;
  for (int d = m.D - 1; d >= 0; d--) {
    const uint32_t id = (uint32_t)(lin % m.shape[d]);
    lin /= m.shape[d];
    const int lv = marks[m.markoff[d] + id];
    level = lv > level ? lv : level;
  }
  return level;
}

// Outlier slots: ONE atomicAdd per workgroup and round of kQuantPerRound elements (the slots of
// the single outlier list come from one address, ~11 ns per atomic once they queue; with one
// atomic per wave a field whose values are mostly out of the dictionary -- D = 5: the quantum
// shrinks with 1 + 3^D -- spent 3.1 of its 4.0 ms per step here: 8 x 8 x 64^3, 262 144 atomics).
constexpr int kQuantEPT = 4;                       // elements per thread and round
constexpr int kQuantPerRound = 256 * kQuantEPT;    // elements per workgroup and round
template <typename T>
__global__ void __launch_bounds__(256)
k_quantize(QuantMeta m, size_t total, const T *__restrict__ v, const int *__restrict__ marks,
           const T *__restrict__ qz, const T *__restrict__ vol, int64_t dict_size,
           int prep_huffman, int64_t *__restrict__ q, unsigned long long *outlier_count,
           uint64_t *__restrict__ outlier_idx, int64_t *__restrict__ outlier_val,
           unsigned long long outlier_cap) {
  __shared__ unsigned wcnt[4];
  __shared__ unsigned long long gbase;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (uniform trip count per workgroup: the barriers below are reached by every thread)
  const size_t stride = (size_t)gridDim.x * kQuantPerRound;
  for (size_t base = (size_t)blockIdx.x * kQuantPerRound; base < total; base += stride) {
    int64_t qd[kQuantEPT];
    bool outl[kQuantEPT];
#pragma unroll
    for (int k = 0; k < kQuantEPT; k++) {
      const size_t lin = base + (size_t)k * 256 + threadIdx.x;
      qd[k] = 0;
      outl[k] = false;
      if (lin < total) {
        const int level = m.calc_vol ? level_of(m, marks, lin) : 0;
        qd[k] = quantize_one(v[lin], qz[level], m.calc_vol ? vol[level] : (T)1);
        if (prep_huffman) {
          qd[k] += dict_size / 2;
          outl[k] = !(qd[k] >= 0 && qd[k] < dict_size);
        }
      }
    }
    if (prep_huffman) {
      // Out-of-dictionary values: ONE slot request per workgroup and round (a request per wave
      // queues on the counter when many values leave the dictionary: 8 x 8 x 64^3 at 1e-3, 3.1 ms
      // instead of 0.2), and none -- one barrier instead of three, no scan -- for a round without any.
      unsigned mine = 0;
#pragma unroll
      for (int k = 0; k < kQuantEPT; k++) mine += outl[k] ? 1u : 0u;
      if (__syncthreads_or((int)mine)) {
        unsigned incl = mine;  // inclusive scan of the counts inside the wave
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned u = __shfl_up(incl, d, 64);
          if (lane >= d) incl += u;
        }
        if (lane == 63) wcnt[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
          unsigned tot = 0;
          for (int w = 0; w < 4; w++) {
            const unsigned c = wcnt[w];
            wcnt[w] = tot;
            tot += c;
          }
          gbase = tot ? atomicAdd(outlier_count, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long o = gbase + wcnt[wave] + (incl - mine);
#pragma unroll
        for (int k = 0; k < kQuantEPT; k++)
          if (outl[k]) {
            if (o < outlier_cap) {
              outlier_idx[o] = base + (size_t)k * 256 + threadIdx.x;
              outlier_val[o] = qd[k];
            }
            qd[k] = 0;
            o++;
          }
        __syncthreads();  // (wcnt / gbase are rewritten by the next round)
      }
    }
#pragma unroll
    for (int k = 0; k < kQuantEPT; k++) {
      const size_t lin = base + (size_t)k * 256 + threadIdx.x;
      if (lin < total) q[lin] = qd[k];
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_dequantize(QuantMeta m, size_t total, const int64_t *__restrict__ q,
             const int *__restrict__ marks, const T *__restrict__ qz, const T *__restrict__ vol,
             int64_t dict_size, int prep_huffman, T *__restrict__ v) {
  for (size_t lin = (size_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
       lin += (size_t)gridDim.x * blockDim.x) {
    const int level = m.calc_vol ? level_of(m, marks, lin) : 0;
    int64_t qd = q[lin];
    if (prep_huffman) qd -= dict_size / 2;
    const T volume = m.calc_vol ? vol[level] : (T)1;
    v[lin] = (qz[level] * volume) * (T)qd;
  }
}

// OutlierRestore (LinearQuantization.hpp:304-350)
__global__ void __launch_bounds__(256)
k_outlier_restore(int64_t *__restrict__ q, uint64_t total, const uint64_t *__restrict__ idx,
                  const int64_t *__restrict__ val, uint64_t count) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (an index outside the array can only come from a damaged stream: skip it, never write
  // out of bounds)
  if (t < count && idx[t] < total) q[idx[t]] = val[t];
}

// The coarse corner box of the reordered layout -- what every level below the finest reads --
// as a compact int64 array: 16-bit symbols widened (k_widen_box), then the outliers that lie
// inside the box written over them (k_outlier_restore_box). Up to four dimensions, leading 1s.
struct BoxMap {
  uint32_t n[4];  // the full array
  uint32_t m[4];  // the box [0, m0) x [0, m1) x [0, m2) x [0, m3)
};
// (one wave per row of the box: the row's position is scalar arithmetic, the lanes stream it)
__global__ void __launch_bounds__(256)
k_widen_box(const uint16_t *__restrict__ sym, int64_t *__restrict__ box, BoxMap B, uint32_t rows) {
  const int lane = threadIdx.x & 63;
  for (uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const uint32_t c = row % B.m[2], ti = row / B.m[2];
    const uint32_t i = ti % B.m[1], t = ti / B.m[1];
    const uint16_t *src = sym + (((size_t)t * B.n[1] + i) * B.n[2] + c) * B.n[3];
    int64_t *dst = box + (size_t)row * B.m[3];
    for (uint32_t f = lane; f < B.m[3]; f += 64) dst[f] = (int64_t)src[f];
  }
}
__global__ void __launch_bounds__(256)
k_outlier_restore_box(int64_t *__restrict__ box, BoxMap B, const uint64_t *__restrict__ idx,
                      const int64_t *__restrict__ val, uint64_t count) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  uint64_t lin = idx[k];
  const uint32_t f = (uint32_t)(lin % B.n[3]);
  lin /= B.n[3];
  const uint32_t c = (uint32_t)(lin % B.n[2]);
  lin /= B.n[2];
  const uint32_t i = (uint32_t)(lin % B.n[1]);
  const uint64_t t = lin / B.n[1];
  // (outside the box: an outlier of the finest level, found through the table; an index outside the
  // array can only come from a damaged stream and falls outside the box as well)
  if (t < B.m[0] && i < B.m[1] && c < B.m[2] && f < B.m[3])
    box[(((size_t)t * B.m[1] + i) * B.m[2] + c) * B.m[3] + f] = val[k];
}

// ---------------------------------------------------------------------------
// config.reorder == 1: the quantized array level by level ("level linearised",
// Quantization/LinearQuantization.hpp:46-146 calc_level_offset + :588-605 slot of a level).
// A pure permutation of the reordered N-D array; one thread per element.
// ---------------------------------------------------------------------------
constexpr int kLinMaxLevels = 40;
struct LinMeta {
  int D, L;
  uint32_t shape[5];
  uint32_t markoff[5];
  uint32_t lshape[kLinMaxLevels + 1][5];  // shape of level l
};

__device__ __forceinline__ uint64_t linearized_position(const LinMeta &m, const int *__restrict__ marks,
                                                        uint64_t lin) {
  uint32_t idx[5];
  int mk[5];
  int level = 0;
  for (int d = m.D - 1; d >= 0; d--) {
    idx[d] = (uint32_t)(lin % m.shape[d]);
    lin /= m.shape[d];
    mk[d] = marks[m.markoff[d] + idx[d]];
    level = mk[d] > level ? mk[d] : level;
  }
  uint64_t stride = 1, cstride = 1, thread_offset = 0, coarse_offset = 0, base = level > 0 ? 1 : 0;
  for (int d = m.D - 1; d >= 0; d--) {
    const uint32_t fine = m.lshape[level][d];
    const uint32_t coarse = level > 0 ? m.lshape[level - 1][d] : 0u;
    const uint32_t bit = mk[d] == level ? 1u : 0u;  // the node is a level-`level` node along d
    const uint32_t r = bit ? idx[d] - coarse : idx[d];
    uint32_t g;  // natural position in the level's fine grid
    if (level == 0) g = r;
    else if (fine % 2 == 0 && r == fine / 2) g = fine - 1;
    else g = r * 2 + bit;
    thread_offset += (uint64_t)g * stride;
    stride *= fine;
    if ((g & 1u) && g != fine - 1) coarse_offset = 0;
    if (g) coarse_offset += (uint64_t)((g - 1) / 2 + 1) * cstride;
    cstride *= fine / 2 + 1;
    if (level > 0) base *= coarse;
  }
  if (level == 0) coarse_offset = 0;
  return base + (thread_offset - coarse_offset);
}

template <typename QT, bool INV>
__global__ void __launch_bounds__(256)
k_level_linearize(LinMeta m, const int *__restrict__ marks, size_t total, const QT *__restrict__ in,
                  QT *__restrict__ out) {
  for (size_t lin = (size_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
       lin += (size_t)gridDim.x * blockDim.x) {
    const uint64_t p = linearized_position(m, marks, lin);
    if (INV) out[lin] = in[p];
    else out[p] = in[lin];
  }
}

// outlier indices (reordered N-D linear index -> linearised position); the count stays on the
// device (capped at `cap`), indices outside the array are left alone
__global__ void __launch_bounds__(256)
k_linearize_indices(LinMeta m, const int *__restrict__ marks, size_t total, uint64_t *__restrict__ idx,
                    const unsigned long long *__restrict__ d_count, unsigned long long count,
                    unsigned long long cap) {
  unsigned long long n = d_count ? *d_count : count;
  n = n < cap ? n : cap;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t lin = idx[i];
    if (lin < total) idx[i] = linearized_position(m, marks, lin);
  }
}

// ---------------------------------------------------------------------------
// Norm reductions (CompressionLowLevel/NormCalculator.hpp:44-71). absmax is
// order independent; the square sum is accumulated per thread / wave / block in
// T and combined with one atomic per block on a double, so it is deterministic
// only up to the atomic order (the reference's GPU path is a tree reduction and
// is not bit-reproducible either).
// ---------------------------------------------------------------------------
template <typename T> struct Vec16 { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

#ifndef MGH_ABSMAX_UNROLL4
#define MGH_ABSMAX_UNROLL4 1
#endif
template <typename T>
__global__ void __launch_bounds__(256)
k_absmax(const T *__restrict__ v, size_t n, unsigned long long *out_bits, size_t n_cold = 0,
         unsigned long long *zero_a = nullptr, unsigned long long *zero_b = nullptr) {
  // (MGH_INLINE_QP: the words k_make_qparams used to reset -- the call's outlier counter, the norm
  // scalar of the NEXT call -- nothing touches them while the norm is reduced)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (zero_a) *zero_a = 0;
    if (zero_b) *zero_b = 0;
  }
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  T m = 0;
  // 16-byte loads over the aligned body (v comes 16-byte aligned from the allocator;
  // a misaligned pointer falls back to the scalar loop entirely)
  const bool aligned = (reinterpret_cast<uintptr_t>(v) & 15) == 0;
  const size_t nvec = aligned ? n / VN : 0;
  const V *vv = reinterpret_cast<const V *>(v);
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  // The leading n_cold elements are read with nontemporal loads: the pass that follows re-reads
  // the input from its END, and what it finds in the 256 MB memory-side cache is what this kernel
  // read last -- the head of a bigger input only flushes it.
  const size_t nvec_cold = aligned ? (n_cold / VN) : 0;
  size_t i0 = tid;
  for (; i0 < nvec_cold; i0 += nth) {
    typedef T NV __attribute__((ext_vector_type(VN)));
    const NV x = __builtin_nontemporal_load(reinterpret_cast<const NV *>(vv + i0));
    const T *xs = reinterpret_cast<const T *>(&x);
#pragma unroll
    for (int u = 0; u < VN; u++) {
      const T a = abs_t(xs[u]);
      m = a > m ? a : m;
    }
  }
  // (four loads in flight per lane: MGH_ABSMAX_UNROLL, A/B in NOTES.md)
  size_t i = i0;
#if MGH_ABSMAX_UNROLL4
  for (; i + 3 * nth < nvec; i += 4 * nth) {
    const V x0 = vv[i], x1 = vv[i + nth], x2 = vv[i + 2 * nth], x3 = vv[i + 3 * nth];
    const T *p0 = reinterpret_cast<const T *>(&x0), *p1 = reinterpret_cast<const T *>(&x1),
            *p2 = reinterpret_cast<const T *>(&x2), *p3 = reinterpret_cast<const T *>(&x3);
#pragma unroll
    for (int u = 0; u < VN; u++) {
      const T a0 = abs_t(p0[u]), a1 = abs_t(p1[u]), a2 = abs_t(p2[u]), a3 = abs_t(p3[u]);
      const T b0 = a0 > a1 ? a0 : a1, b1 = a2 > a3 ? a2 : a3;
      const T c = b0 > b1 ? b0 : b1;
      m = c > m ? c : m;
    }
  }
#endif
  for (; i < nvec; i += nth) {
    const V x = vv[i];
    const T *xs = reinterpret_cast<const T *>(&x);
#pragma unroll
    for (int u = 0; u < VN; u++) {
      const T a = abs_t(xs[u]);
      m = a > m ? a : m;
    }
  }
  for (size_t k = nvec * VN + tid; k < n; k += nth) {
    const T a = abs_t(v[k]);
    m = a > m ? a : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const T o = __shfl_down(m, off, 64);
    m = o > m ? o : m;
  }
  __shared__ T sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = sm[w] > m ? sm[w] : m;
    // non-negative IEEE values order like their bit patterns
    unsigned long long bits;
    if (sizeof(T) == 4) bits = __float_as_uint((float)m); else bits = __double_as_longlong((double)m);
    atomicMax(out_bits, bits);
  }
}

// Pure stream with the read/write mix of the top-level pass (mgh_stream_calibrate).
template <typename T>
__global__ void __launch_bounds__(256)
k_stream_mix(const T *__restrict__ in, int64_t *__restrict__ out, T *__restrict__ s1, T *__restrict__ s2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T x = in[i];
    __builtin_nontemporal_store((int64_t)(int)x, &out[i]);
    if ((i & 7) == 0) {
      s1[i >> 3] = x;
      s2[i >> 3] = x + (T)1;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_sqsum(const T *__restrict__ v, size_t n, double *out, size_t n_cold = 0,
        unsigned long long *zero_a = nullptr, unsigned long long *zero_b = nullptr) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // (see k_absmax)
    if (zero_a) *zero_a = 0;
    if (zero_b) *zero_b = 0;
  }
  // 16-byte loads over the aligned body like k_absmax (the L2 norm is a sum: its last bits depend
  // on the order of the additions -- across lanes, waves and the atomicAdd below -- in any case);
  // the leading n_cold elements with nontemporal loads
  constexpr int VN = Vec16<T>::N;
  typedef T NV __attribute__((ext_vector_type(VN)));
  T acc = 0;
  const bool aligned = (reinterpret_cast<uintptr_t>(v) & 15) == 0;
  const size_t nvec = aligned ? n / VN : 0, nvec_cold = aligned ? n_cold / VN : 0;
  const NV *vv = reinterpret_cast<const NV *>(v);
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  // (four loads in flight per lane as in k_absmax: measured SLOWER here, 196 -> 224 us at 512^3 f64)
  for (size_t i = tid; i < nvec; i += nth) {
    const NV x = i < nvec_cold ? __builtin_nontemporal_load(vv + i) : vv[i];
#pragma unroll
    for (int u = 0; u < VN; u++) acc += x[u] * x[u];
  }
  for (size_t k = nvec * VN + tid; k < n; k += nth) acc += v[k] * v[k];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ T sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) acc += sm[w];
    atomicAdd(out, (double)acc);
  }
}

} // namespace mgh
