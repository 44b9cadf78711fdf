// Generic N-D (D <= 5) kernels: first correct HIP path for D = 4, 5 (one thread per element /
// pencil, no tiling yet). Semantics = the reference's ND drivers
//   CalcCoefficientsND.hpp:25-236  (interpolants built fastest dim first, coefficient =
//                                   original - interpolant),
//   CalcCorrectionND.hpp:25-267    (mass/restriction sweeps along D-1..0, the first one
//                                   reading the all-coarse corner as zero; Thomas solves in
//                                   the same order),
//   CoefficientsRestoreND.hpp      (inverse),
// with the same element arithmetic as kernels_v1.hpp (bit-identical for D <= 3, which the
// tests use as a cross-check).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_v1.hpp"

namespace mgh {

constexpr int kNd = 5;

struct NdBox {
  int D;
  uint32_t n[kNd];  // fine extents of the level
  uint32_t m[kNd];  // coarse extents
  uint64_t fs[kNd];  // element strides of the full (reordered) array
};

// the level's fine box covers the whole (dense) array: box-order = array-order
inline bool nd_box_is_whole_array(const NdBox &b) {
  uint64_t st = 1;
  for (int d = b.D - 1; d >= 0; d--) {
    if (b.fs[d] != st) return false;
    st *= b.n[d];
  }
  return true;
}

template <typename T> struct NdTables {
  const T *ratio[kNd];
};

__device__ __forceinline__ void nd_unravel(int D, const uint32_t *e, uint64_t lin, uint32_t *idx) {
  if (lin < (1ull << 32)) {  // 32-bit divisions are several times cheaper
    uint32_t l32 = (uint32_t)lin;
    for (int d = D - 1; d >= 0; d--) {
      const uint32_t q = l32 / e[d];
      idx[d] = l32 - q * e[d];
      l32 = q;
    }
    return;
  }
  for (int d = D - 1; d >= 0; d--) {
    idx[d] = (uint32_t)(lin % e[d]);
    lin /= e[d];
  }
}
__device__ __forceinline__ uint64_t nd_ravel(int D, const uint32_t *e, const uint32_t *idx) {
  uint64_t lin = 0;
  for (int d = 0; d < D; d++) lin = lin * e[d] + idx[d];
  return lin;
}

// compact natural-order copy of the fine box out of the full array (CopyND)
template <typename T>
__global__ void __launch_bounds__(256)
k_nd_gather(NdBox b, const T *__restrict__ v, T *__restrict__ w, uint64_t total, int scatter) {
  for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
       lin += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t idx[kNd];
    nd_unravel(b.D, b.n, lin, idx);
    uint64_t off = 0;
    for (int d = 0; d < b.D; d++) off += idx[d] * b.fs[d];
    if (scatter) const_cast<T *>(v)[off] = w[lin]; else w[lin] = v[off];
  }
}

// Nested lerps over K odd dims, slowest dim outermost (od[0] = fastest odd dim is the innermost
// lerp); fully inlined recursion: no local arrays.
template <typename T, int K> struct NdLerp {
  static __device__ __forceinline__ T run(const T *w, uint64_t off, const uint64_t *st,
                                          const T *t) {
    return lerp_ref(NdLerp<T, K - 1>::run(w, off - st[K - 1], st, t),
                    NdLerp<T, K - 1>::run(w, off + st[K - 1], st, t), t[K - 1]);
  }
};
template <typename T> struct NdLerp<T, 0> {
  static __device__ __forceinline__ T run(const T *w, uint64_t off, const uint64_t *, const T *) {
    return w[off];
  }
};

// interpolant of the node at natural position pos (linear offset wl in the natural-order
// compact box w with strides ns; odd dims marked): nested lerps, fastest dim innermost
template <typename T>
__device__ __forceinline__ T nd_interp(const NdBox &b, const NdTables<T> &tb, const T *w,
                                       const uint32_t *pos, const bool *odd, uint64_t wl,
                                       const uint64_t *ns) {
  uint64_t st[kNd];
  T t[kNd];
  int nod = 0;
  for (int d = b.D - 1; d >= 0; d--)
    if (odd[d]) {
      st[nod] = ns[d];
      t[nod] = tb.ratio[d][pos[d] - 1];
      nod++;
    }
  switch (nod) {
  case 1: return NdLerp<T, 1>::run(w, wl, st, t);
  case 2: return NdLerp<T, 2>::run(w, wl, st, t);
  case 3: return NdLerp<T, 3>::run(w, wl, st, t);
  case 4: return NdLerp<T, 4>::run(w, wl, st, t);
  case 5: return NdLerp<T, 5>::run(w, wl, st, t);
  default: return w[wl];
  }
}

// mode 0: decompose: v[reordered] = node - interpolant (coarse nodes copied)
// mode 1: recompose pass A: w[natural] = v[reordered] for all-coarse nodes
// mode 2: recompose pass B: w[natural] = v[reordered] + interpolant(w) for the other nodes
template <typename T>
__global__ void __launch_bounds__(256)
k_nd_coeff(NdBox b, NdTables<T> tb, T *__restrict__ w, T *__restrict__ v, uint64_t total,
           int mode) {
  for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
       lin += (uint64_t)gridDim.x * blockDim.x) {
    // threads run over the NATURAL order of the fine box: the centre and its neighbours -- on
    // average five reads per node -- are then contiguous across a wave; the one access to the
    // reordered array splits into the coarse and the coefficient half of a row
    uint32_t idx[kNd], pos[kNd];
    bool odd[kNd], any = false;
    nd_unravel(b.D, b.n, lin, pos);
    uint64_t ns[kNd];  // strides of the natural-order compact box
    {
      uint64_t sacc = 1;
      for (int d = b.D - 1; d >= 0; d--) {
        ns[d] = sacc;
        sacc *= b.n[d];
      }
    }
    uint64_t off = 0;
    const uint64_t wl = lin;
    for (int d = 0; d < b.D; d++) {
      const uint32_t p = pos[d], n = b.n[d], m = b.m[d];
      odd[d] = (p & 1) && !(n % 2 == 0 && p == n - 1);  // (inverse of fine_pos)
      idx[d] = odd[d] ? m + (p - 1) / 2 : (p == n - 1 ? m - 1 : p / 2);
      any |= odd[d];
      off += idx[d] * b.fs[d];
    }
    if (mode == 0) {
      const T centre = w[wl];
      v[off] = any ? centre - nd_interp<T>(b, tb, w, pos, odd, wl, ns) : centre;
    } else if (mode == 1) {
      if (!any) w[wl] = v[off];
    } else if (any) {
      T res = v[off];
      res += nd_interp<T>(b, tb, w, pos, odd, wl, ns);
      w[wl] = res;
    }
  }
}

// mass/restriction sweep along dim a. in: extents e, strides is; out: compact, extents e with
// e[a] -> m. One thread per output element.
struct NdSweep {
  int D, a;
  uint32_t e[kNd];   // input extents
  uint64_t is[kNd];  // input strides
  uint32_t mc[kNd];  // coarse extents (zero rule)
  uint32_t n, m;     // fine / coarse size of dim a
  int zero_all_coarse;
};

template <typename T>
__global__ void __launch_bounds__(256)
k_nd_lpk(NdSweep s, const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ mt,
         uint64_t total) {
  for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
       lin += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t eo[kNd], idx[kNd];
    for (int d = 0; d < s.D; d++) eo[d] = s.e[d];
    eo[s.a] = s.m;
    nd_unravel(s.D, eo, lin, idx);
    uint64_t base = 0;
    bool ez = s.zero_all_coarse;
    for (int d = 0; d < s.D; d++) {
      if (d == s.a) continue;
      base += idx[d] * s.is[d];
      if (idx[d] >= s.mc[d]) ez = false;
    }
    const uint32_t q = idx[s.a], m = s.m, nodd = s.n - s.m;
    const uint64_t st = s.is[s.a];
    const T *pe = in + base, *po = in + base + m * st;
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
    out[lin] = tc;
  }
}

// Thomas solve along dim a of a compact array with extents e; one thread per pencil
template <typename T>
__global__ void __launch_bounds__(256)
k_nd_ipk(int D, int a, NdSweep s, T *__restrict__ x, const T *__restrict__ tt, uint64_t npencil) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npencil;
       p += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t el[kNd], idx[kNd];
    for (int d = 0; d < D; d++) el[d] = s.e[d];
    el[a] = 1;
    nd_unravel(D, el, p, idx);
    uint64_t base = 0;
    for (int d = 0; d < D; d++) base += idx[d] * s.is[d];
    const uint64_t st = s.is[a];
    const uint32_t n = s.e[a];
    T *q = x + base;
    T prev = 0;
    for (uint32_t i = 0; i < n; i++) {
      T cur = q[i * st];
      cur = cur - prev * tt[i];
      q[i * st] = cur;
      prev = cur;
    }
    prev = 0;
    for (uint32_t kk = 0; kk < n; kk++) {
      const uint32_t i = n - 1 - kk;
      T cur = q[i * st];
      cur = (cur - tt[n + i] * prev) / tt[2 * n + i];
      q[i * st] = cur;
      prev = cur;
    }
  }
}

// ---- row-wise variants (round 6) ---------------------------------------------------------------
// The one-thread-per-element kernels above pay five integer divisions per element for its position
// (no hardware division: ~25 instructions each) and decide every branch per lane. Here a wave owns
// kNdRowsPerWave consecutive ROWS (a row = all dimensions but the fastest fixed): the position of the
// first row is unravelled once, the following ones by counting up; parity, reordered offset and
// interpolation partners of the slow dimensions are the same for the whole row (wave-uniform
// branches), the lanes run along the fastest dimension. Same element arithmetic in the same order
// (nested lerps slowest dimension outermost; mass_apply of k_nd_lpk): bit-identical results.
// Dimensions are right-aligned to kNd = 5 (leading extents 1).
constexpr int kNdRowsPerWave = 8;

struct NdRowBox {
  uint32_t n[kNd], m[kNd];   // fine / coarse extents, right-aligned
  uint64_t fs[kNd];          // strides of the reordered (full) array
  uint64_t ns[kNd];          // strides of the natural-order compact box
  uint64_t rows;             // product of n[0..3]
};

// position of the row `row` in dims 0..3 (extents e), then the next rows by odometer steps
__device__ __forceinline__ void nd_row_start(const uint32_t *e, uint64_t row, uint32_t *pos) {
  if (row < (1ull << 32)) {
    uint32_t r = (uint32_t)row;
    for (int d = kNd - 2; d >= 0; d--) {
      const uint32_t q = r / e[d];
      pos[d] = r - q * e[d];
      r = q;
    }
  } else {
    for (int d = kNd - 2; d >= 0; d--) {
      pos[d] = (uint32_t)(row % e[d]);
      row /= e[d];
    }
  }
}
__device__ __forceinline__ void nd_row_next(const uint32_t *e, uint32_t *pos) {
  for (int d = kNd - 2; d >= 0; d--) {
    if (++pos[d] < e[d]) return;
    pos[d] = 0;
  }
}

// modes as in k_nd_coeff
template <typename T>
__global__ void __launch_bounds__(256)
k_nd_coeff_rows(NdRowBox b, NdTables<T> tb, T *__restrict__ w, T *__restrict__ v, int mode) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (uint64_t)blockIdx.x * 4;
  const uint64_t nwave = (uint64_t)gridDim.x * 4;
  const uint32_t nf = b.n[kNd - 1], mf = b.m[kNd - 1];
  const T *__restrict__ rf = tb.ratio[kNd - 1];
  for (uint64_t r0 = wave * kNdRowsPerWave; r0 < b.rows; r0 += nwave * kNdRowsPerWave) {
    uint32_t pos[kNd - 1];
    nd_row_start(b.n, r0, pos);
    const uint64_t r1 = min(b.rows, r0 + kNdRowsPerWave);
    for (uint64_t row = r0; row < r1; row++) {
      // the row's slow dimensions: parity, reordered offset, natural offset, interpolation partners
      uint64_t off_slow = 0, wl_row = 0;
      uint64_t s1 = 0, s2 = 0, s3 = 0, s4 = 0;  // natural strides of the odd slow dims, fastest first
      T t1 = 0, t2 = 0, t3 = 0, t4 = 0;
      int nod = 0;
#pragma unroll
      for (int d = kNd - 2; d >= 0; d--) {
        const uint32_t p = pos[d], n = b.n[d], m = b.m[d];
        const bool odd = (p & 1) && !(n % 2 == 0 && p == n - 1);
        const uint32_t idx = odd ? m + (p - 1) / 2 : (p == n - 1 ? m - 1 : p / 2);
        off_slow += idx * b.fs[d];
        wl_row += p * b.ns[d];
        if (odd) {
          const uint64_t st = b.ns[d];
          const T t = tb.ratio[d][p - 1];
          if (nod == 0) s1 = st, t1 = t;
          else if (nod == 1) s2 = st, t2 = t;
          else if (nod == 2) s3 = st, t3 = t;
          else s4 = st, t4 = t;
          nod++;
        }
      }
      const uint64_t sa[kNd] = {1, s1, s2, s3, s4};  // with the fastest dimension in front ...
      const uint64_t sb[kNd] = {s1, s2, s3, s4, 0};  // ... and without it
      for (uint32_t p = lane; p < nf; p += 64) {
        const bool odd_f = (p & 1) && !(nf % 2 == 0 && p == nf - 1);
        const uint32_t idx_f = odd_f ? mf + (p - 1) / 2 : (p == nf - 1 ? mf - 1 : p / 2);
        const uint64_t off = off_slow + idx_f * b.fs[kNd - 1], wl = wl_row + p;
        const bool any = odd_f || nod > 0;
        if (mode == 1) {
          if (!any) w[wl] = v[off];
          continue;
        }
        if (mode == 2 && !any) continue;
        T interp = 0;
        if (any) {
          if (odd_f) {
            const T ta[kNd] = {rf[p - 1], t1, t2, t3, t4};
            switch (nod) {
            case 0: interp = NdLerp<T, 1>::run(w, wl, sa, ta); break;
            case 1: interp = NdLerp<T, 2>::run(w, wl, sa, ta); break;
            case 2: interp = NdLerp<T, 3>::run(w, wl, sa, ta); break;
            case 3: interp = NdLerp<T, 4>::run(w, wl, sa, ta); break;
            default: interp = NdLerp<T, 5>::run(w, wl, sa, ta); break;
            }
          } else {
            const T tb2[kNd] = {t1, t2, t3, t4, 0};
            switch (nod) {
            case 1: interp = NdLerp<T, 1>::run(w, wl, sb, tb2); break;
            case 2: interp = NdLerp<T, 2>::run(w, wl, sb, tb2); break;
            case 3: interp = NdLerp<T, 3>::run(w, wl, sb, tb2); break;
            default: interp = NdLerp<T, 4>::run(w, wl, sb, tb2); break;
            }
          }
        }
        if (mode == 0) {
          const T centre = w[wl];
          v[off] = any ? centre - interp : centre;
        } else {
          T res = v[off];
          res += interp;
          w[wl] = res;
        }
      }
      nd_row_next(b.n, pos);
    }
  }
}

// k_nd_lpk by rows of the OUTPUT (extents eo = e with e[a] -> m, right-aligned, a in 0..4)
struct NdRowSweep {
  int a;
  uint32_t eo[kNd];   // output extents
  uint64_t is[kNd];   // input strides
  uint32_t mc[kNd];   // coarse extents (zero rule)
  uint32_t n, m;      // fine / coarse size of dim a
  int zero_all_coarse;
  uint64_t rows;      // product of eo[0..3]
};

template <typename T>
__global__ void __launch_bounds__(256)
k_nd_lpk_rows(NdRowSweep s, const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ mt) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (uint64_t)blockIdx.x * 4;
  const uint64_t nwave = (uint64_t)gridDim.x * 4;
  const uint32_t ef = s.eo[kNd - 1], m = s.m, nodd = s.n - s.m;
  const uint64_t st = s.is[s.a];
  for (uint64_t r0 = wave * kNdRowsPerWave; r0 < s.rows; r0 += nwave * kNdRowsPerWave) {
    uint32_t pos[kNd - 1];
    nd_row_start(s.eo, r0, pos);
    const uint64_t r1 = min(s.rows, r0 + kNdRowsPerWave);
    for (uint64_t row = r0; row < r1; row++) {
      uint64_t base = 0;
      bool ez_row = s.zero_all_coarse;
#pragma unroll
      for (int d = 0; d < kNd - 1; d++) {
        if (d == s.a) continue;
        base += pos[d] * s.is[d];
        if (pos[d] >= s.mc[d]) ez_row = false;
      }
      T *o = out + row * ef;
      if (s.a == kNd - 1) {
        // along the fastest dimension: lane = output index q
        const T *pe = in + base, *po = in + base + m * st;
        for (uint32_t q = lane; q < ef; q += 64) {
          const bool ez = ez_row;
          const T a = (q >= 1 && !ez) ? pe[(q - 1) * st] : (T)0;
          const T bq = (q >= 1 && q - 1 < nodd) ? po[(q - 1) * st] : (T)0;
          const T c = ez ? (T)0 : pe[q * st];
          const T d = (q < nodd) ? po[q * st] : (T)0;
          const T e = (q + 1 < m && !ez) ? pe[(q + 1) * st] : (T)0;
          const T w0 = mt[0 * m + q], w1 = mt[1 * m + q], w2 = mt[2 * m + q], w3 = mt[3 * m + q],
                  w4 = mt[4 * m + q], w5 = mt[5 * m + q], w6 = mt[6 * m + q], r1w = mt[7 * m + q],
                  r4w = mt[8 * m + q];
          const T tb = a * w0 + bq * w1 + c * w2;
          T tc = bq * w2 + c * w3 + d * w4;
          const T td = c * w4 + d * w5 + e * w6;
          tc += tb * r1w + td * r4w;
          o[q] = tc;
        }
      } else {
        // along a slow dimension: q is the row's, the lanes run along the fastest dimension
        const uint32_t q = pos[s.a];
        const T w0 = mt[0 * m + q], w1 = mt[1 * m + q], w2 = mt[2 * m + q], w3 = mt[3 * m + q],
                w4 = mt[4 * m + q], w5 = mt[5 * m + q], w6 = mt[6 * m + q], r1w = mt[7 * m + q],
                r4w = mt[8 * m + q];
        const uint64_t sf = s.is[kNd - 1];
        for (uint32_t p = lane; p < ef; p += 64) {
          const bool ez = ez_row && p < s.mc[kNd - 1];
          const T *pe = in + base + p * sf, *po = pe + m * st;
          const T a = (q >= 1 && !ez) ? pe[(q - 1) * st] : (T)0;
          const T bq = (q >= 1 && q - 1 < nodd) ? po[(q - 1) * st] : (T)0;
          const T c = ez ? (T)0 : pe[q * st];
          const T d = (q < nodd) ? po[q * st] : (T)0;
          const T e = (q + 1 < m && !ez) ? pe[(q + 1) * st] : (T)0;
          const T tb = a * w0 + bq * w1 + c * w2;
          T tc = bq * w2 + c * w3 + d * w4;
          const T td = c * w4 + d * w5 + e * w6;
          tc += tb * r1w + td * r4w;
          o[p] = tc;
        }
      }
      nd_row_next(s.eo, pos);
    }
  }
}

// v[coarse box, full strides] +/-= corr[compact]
template <typename T>
__global__ void __launch_bounds__(256)
k_nd_apply(NdBox b, const T *__restrict__ corr, T *__restrict__ v, uint64_t total, int sign) {
  for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
       lin += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t idx[kNd];
    nd_unravel(b.D, b.m, lin, idx);
    uint64_t off = 0;
    for (int d = 0; d < b.D; d++) off += idx[d] * b.fs[d];
    if (sign > 0) v[off] += corr[lin]; else v[off] -= corr[lin];
  }
}

} // namespace mgh
