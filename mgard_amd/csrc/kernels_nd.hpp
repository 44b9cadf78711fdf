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

// The row-wise kernel's form: the fastest dimension innermost as in NdLerp with it in front, but
// every corner row of the slow dimensions is LOADED ONCE -- each lane one element -- and the odd
// lanes take their two partners p - 1 and p + 1 from the neighbouring lanes' registers (one DPP
// move each) instead of loading the corner row three times (p - 1 and p + 1 for the odd lanes, p
// for the even ones).
__device__ __forceinline__ int lane_below(int x) {  // lane i: the value of lane i - 1 (wave_shr:1)
  return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int lane_above(int x) {  // lane i: the value of lane i + 1 (wave_shl:1)
  return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ float lane_below(float x) { return __int_as_float(lane_below(__float_as_int(x))); }
__device__ __forceinline__ float lane_above(float x) { return __int_as_float(lane_above(__float_as_int(x))); }
__device__ __forceinline__ double lane_below(double x) {
  return __hiloint2double(lane_below(__double2hiint(x)), lane_below(__double2loint(x)));
}
__device__ __forceinline__ double lane_above(double x) {
  return __hiloint2double(lane_above(__double2hiint(x)), lane_above(__double2loint(x)));
}
// (a wave-uniform pointer and the lane's BYTE offset in 32 bits: base[p] with a 32-bit p is p * 4 in
// 64 bits to the compiler)
template <typename T> __device__ __forceinline__ T *at32(T *base, uint32_t byte_off) {
  return (T *)((char *)base + byte_off);
}
template <typename T> __device__ __forceinline__ const T *at32(const T *base, uint32_t byte_off) {
  return (const T *)((const char *)base + byte_off);
}
// corner: the corner row (wave-uniform); lb: the byte offset of the element the lane loads of it.
// The loads are unconditional -- a load behind a condition is a branch and a wait PER CORNER, the
// rows' corners then come one round trip after the other -- so lb must be a valid element for
// every lane (k_nd_coeff_rows chooses it).
template <typename T, int K> struct NdLerpRow {
  static __device__ __forceinline__ T run(const T *corner, uint32_t lb, bool odd_f, T tf, const uint32_t *st,
                                          const T *t) {
    return lerp_ref(NdLerpRow<T, K - 1>::run(corner - st[K - 1], lb, odd_f, tf, st, t),
                    NdLerpRow<T, K - 1>::run(corner + st[K - 1], lb, odd_f, tf, st, t), t[K - 1]);
  }
};
template <typename T> struct NdLerpRow<T, 0> {
  static __device__ __forceinline__ T run(const T *corner, uint32_t lb, bool odd_f, T tf, const uint32_t *,
                                          const T *) {
    const T c = *at32(corner, lb);
    const T lo = lane_below(c), hi = lane_above(c);
    return odd_f ? lerp_ref(lo, hi, tf) : c;
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
// (no hardware division: ~25 instructions each) and decide every branch per lane. Here the lanes run
// along the fastest dimension and a wave owns ROWS (a row = all dimensions but the fastest fixed):
// parity, reordered offset and interpolation partners of the slow dimensions are the same for the
// whole row (wave-uniform branches). Same element arithmetic in the same order (nested lerps
// slowest dimension outermost; mass_apply of k_nd_lpk): bit-identical results. Dimensions are
// right-aligned to kNd = 5 (leading extents 1). Which kernel runs what (capi.hip):
//   coefficients            k_nd_coeff_rows  groups of up to kNdRowsPerWave rows of one dimension-3 line
//   first correction sweep  k_nd_lpk_fast    the same groups, lane = output index
//   the other sweeps        k_nd_lpk_mid     a 3-D view of the compact array, one element a thread
//   (k_nd_lpk_rows: the sweeps of round 6's first version -- kNdRowsPerWave consecutive rows by
//   odometer steps -- left for arrays whose group / tile counts do not fit 32 bits)
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

// The coefficient kernel's view: 32-bit strides (the host sends larger arrays to k_nd_coeff) and the
// rows in GROUPS -- gsz consecutive positions of dimension 3 at one position of dimensions 0..2 --
// so that everything dimensions 0..2 decide (which of them interpolate, with what strides and
// ratios) is worked out once for the group's rows.
struct NdCoeffBox {
  uint32_t n[kNd], m[kNd];   // fine / coarse extents, right-aligned
  uint32_t fs[kNd];          // strides of the reordered (full) array
  uint32_t ns[kNd];          // strides of the natural-order compact box
  uint32_t gsz;              // rows of a group: 2, 4 or 8 (even: a group starts on a coarse position)
  uint32_t gpl;              // groups per line of dimension 3
  uint32_t groups;           // n[0] n[1] n[2] gpl
};

// The rows of one parity class of dimension 3 in a group (odd3: the rows that interpolate along it):
// K slow dimensions interpolate, strides st / ratios tt fastest first (for the odd class entry 0 is
// dimension 3's, its ratio the row's). Everything along the row itself -- the lane's byte offsets,
// the ratio tf -- is the caller's: the same for all rows.
template <typename T, int K> struct NdCoeffClass {
  const NdCoeffBox &b;
  const T *__restrict__ ratio3;
  T *__restrict__ w, *__restrict__ v;
  int mode;
  bool odd3;
  uint32_t off012, wl012;
  const uint32_t *st;
  T tt[K ? K : 1];
  uint32_t pb, fb;   // the lane's BYTE offsets along the row: natural order / reordered
  uint32_t lb;       // ... and of the element it loads of every corner row
  bool act, any, odd_f;
  T tf;

  // the value row q stores (mode 0: into v, else into w; *vr / *wr: the row there, wave-uniform)
  __device__ __forceinline__ T value(uint32_t q, T **vr, T **wr) {
    const uint32_t n3 = b.n[3], m3 = b.m[3];
    const uint32_t idx3 = odd3 ? m3 + (q - 1) / 2 : (q == n3 - 1 ? m3 - 1 : q / 2);
    *vr = v + (off012 + idx3 * b.fs[3]);
    *wr = w + (wl012 + q * b.ns[3]);
    if (odd3) tt[0] = ratio3[q - 1];
    const T mine = act ? (mode == 0 ? *at32(*wr, pb) : *at32(*vr, fb)) : (T)0;  // (in flight with the partners)
    const T interp = NdLerpRow<T, K>::run(*wr, lb, odd_f, tf, st, tt);
    return mode == 0 ? (any ? mine - interp : mine) : mine + interp;
  }
  __device__ __forceinline__ void store(T x, T *vr, T *wr) {
    if (act) {
      if (mode == 0) *at32(vr, fb) = x;
      else *at32(wr, pb) = x;
    }
  }
  // the class's rows of q0 <= q < q1, two at a time: the second row's loads leave before the first
  // row's arithmetic (a row is one round trip to memory; the wave has nothing else to do meanwhile)
  __device__ __forceinline__ void run(uint32_t q0, uint32_t q1) {
    const uint32_t n3 = b.n[3];
    const bool tail = n3 % 2 == 0 && q1 == n3;          // the last position of an even extent: coarse
    const uint32_t qa = odd3 ? q0 + 1 : q0;              // (q0 is a multiple of the group size: even)
    const uint32_t qe = odd3 && tail ? q1 - 1 : q1;
    uint32_t q = qa;
    for (; q + 2 < qe; q += 4) {
      T *o0, *w0, *o1, *w1;
      const T x0 = value(q, &o0, &w0), x1 = value(q + 2, &o1, &w1);
      store(x0, o0, w0);
      store(x1, o1, w1);
    }
    if (q < qe) {
      T *o0, *w0;
      const T x0 = value(q, &o0, &w0);
      store(x0, o0, w0);
    }
    if (!odd3 && tail) {
      T *o0, *w0;
      const T x0 = value(n3 - 1, &o0, &w0);
      store(x0, o0, w0);
    }
  }
};

template <typename T, int K>
__device__ __forceinline__ void nd_coeff_class(const NdCoeffBox &b, const T *__restrict__ ratio3, T *__restrict__ w,
                                               T *__restrict__ v, int mode, uint32_t q0, uint32_t q1, bool odd3,
                                               uint32_t off012, uint32_t wl012, const uint32_t *st, const T *t,
                                               uint32_t p, uint32_t off_f, uint32_t pl, bool own, bool odd_f, T tf) {
  const bool any = odd_f || K > 0;
  NdCoeffClass<T, K> c{b, ratio3, w, v, mode, odd3, off012, wl012, st, {}, p * (uint32_t)sizeof(T),
                       off_f * (uint32_t)sizeof(T), pl * (uint32_t)sizeof(T), own && (mode == 0 || any), any, odd_f, tf};
#pragma unroll
  for (int k = 0; k < K; k++) c.tt[k] = t[k];
  c.run(q0, q1);
}

// modes as in k_nd_coeff
template <typename T>
__global__ void __launch_bounds__(256)
k_nd_coeff_rows(NdCoeffBox b, NdTables<T> tb, T *__restrict__ w, T *__restrict__ v, int mode) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + blockIdx.x * 4;
  const uint32_t nwave = gridDim.x * 4;
  const uint32_t nf = b.n[kNd - 1], mf = b.m[kNd - 1];
  const T *__restrict__ rf = tb.ratio[kNd - 1];
  for (uint32_t g = wave; g < b.groups; g += nwave) {
    // the group: rows q0..q1 of the line at (pos[0], pos[1], pos[2])
    uint32_t line = g / b.gpl;
    const uint32_t q0 = (g - line * b.gpl) * b.gsz, q1 = min(b.n[3], q0 + b.gsz);
    uint32_t pos[3];
#pragma unroll
    for (int d = 2; d >= 0; d--) {
      const uint32_t qq = line / b.n[d];
      pos[d] = line - qq * b.n[d];
      line = qq;
    }
    // dimensions 0..2: reordered offset, natural offset, and of the odd ones stride and ratio, fastest first
    uint32_t off012 = 0, wl012 = 0;
    uint32_t sx[3] = {0, 0, 0};
    T tx[3] = {0, 0, 0};
    int nod = 0;
#pragma unroll
    for (int d = 2; d >= 0; d--) {
      const uint32_t pp = pos[d], n = b.n[d], m = b.m[d];
      const bool odd = (pp & 1) && !(n % 2 == 0 && pp == n - 1);
      const uint32_t idx = odd ? m + (pp - 1) / 2 : (pp == n - 1 ? m - 1 : pp / 2);
      off012 += idx * b.fs[d];
      wl012 += pp * b.ns[d];
      if (odd) {
        const T t = tb.ratio[d][pp - 1];
        if (nod == 0) sx[0] = b.ns[d], tx[0] = t;
        else if (nod == 1) sx[1] = b.ns[d], tx[1] = t;
        else sx[2] = b.ns[d], tx[2] = t;
        nod++;
      }
    }
    const uint32_t so[4] = {b.ns[3], sx[0], sx[1], sx[2]};  // the rows odd in dimension 3: it comes first
    const T to[4] = {0, tx[0], tx[1], tx[2]};
    // (the whole wave goes round, lanes trade values; of a row longer than the wave 62 elements a
    // round, lanes 62 and 63 there only as partners: the odd lane 63 would want element p + 1)
    for (uint32_t p0 = 0, step; p0 < nf; p0 += step) {
      step = nf - p0 <= 64 ? 64 : 62;
      const uint32_t p = p0 + lane;
      const bool in = p < nf, own = in && lane < step;
      const bool odd_f = in && (p & 1) && !(nf % 2 == 0 && p == nf - 1);
      const uint32_t idx_f = odd_f ? mf + (p - 1) / 2 : (p == nf - 1 ? mf - 1 : p / 2);
      const uint32_t off_f = idx_f * b.fs[kNd - 1];
      // every lane loads, of every corner row, an element that is there and coarse along the row:
      // its own, the odd lanes (who only want their neighbours' values) the one before, the lanes
      // beyond the row its last
      const uint32_t pl = !in ? nf - 1 : odd_f ? p - 1 : p;
      if (mode == 1) {  // the all-coarse nodes: even rows of a group with no odd dimension, even lanes
        if (nod == 0 && own && !odd_f)
          for (uint32_t q = q0; q < q1; q++) {
            const bool o = (q & 1) && !(b.n[3] % 2 == 0 && q == b.n[3] - 1);
            if (o) continue;
            const uint32_t idx3 = q == b.n[3] - 1 ? b.m[3] - 1 : q / 2;
            w[wl012 + q * b.ns[3] + p] = v[off012 + idx3 * b.fs[3] + off_f];
          }
        continue;
      }
      const T tf = odd_f ? rf[p - 1] : (T)0;
#define MGH_ND_CLASSES(K)                                                                                              \
  nd_coeff_class<T, K>(b, tb.ratio[3], w, v, mode, q0, q1, false, off012, wl012, sx, tx, p, off_f, pl, own, odd_f, tf); \
  nd_coeff_class<T, K + 1>(b, tb.ratio[3], w, v, mode, q0, q1, true, off012, wl012, so, to, p, off_f, pl, own, odd_f, tf);
      switch (nod) {  // (the group's)
      case 0: MGH_ND_CLASSES(0) break;
      case 1: MGH_ND_CLASSES(1) break;
      case 2: MGH_ND_CLASSES(2) break;
      default: MGH_ND_CLASSES(3) break;
      }
#undef MGH_ND_CLASSES
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

// k_nd_lpk along a dimension that is NOT the fastest: every sweep but the first reads the compact
// result of the one before -- (outer, n along a, inner) with the inner elements contiguous, no zero
// rule -- and that is all the kernel needs to know: no positions in five dimensions, no rows of 33
// elements in 64 lanes. One element a thread, four a thread along the (q, i) plane of one outer
// index; every load unconditional, of a row clamped into the pencil, the value then chosen or not
// (a load behind a per-lane condition is a branch and a wait of its own).
struct NdMidSweep {
  uint32_t outer, n, m, inner;
  uint32_t plane;   // m * inner: outputs per outer index
  uint32_t tiles;   // ceil(plane / 1024)
};

template <typename T>
__global__ void __launch_bounds__(256)
k_nd_lpk_mid(NdMidSweep s, const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ mt) {
  const uint32_t m = s.m, nodd = s.n - s.m, I = s.inner;
  const uint32_t olast = nodd ? nodd - 1 : 0;
  const uint32_t nblk = s.outer * s.tiles;
  for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const uint32_t o = blk / s.tiles, tile = blk - o * s.tiles;
    const T *pe = in + (uint64_t)o * s.n * I;        // the coarse rows of the pencil's plane ...
    const T *po = nodd ? pe + (uint64_t)m * I : pe;  // ... and the odd ones behind them
    T *po_out = out + (uint64_t)o * s.plane;
    uint32_t r = tile * 1024 + threadIdx.x;
    uint32_t q = r / I, i = r - q * I;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (r < s.plane) {
        const uint32_t qm = q >= 1 ? q - 1 : 0;
        const T la = pe[qm * I + i], lb = po[min(qm, olast) * I + i], lc = pe[q * I + i],
                ld = po[min(q, olast) * I + i], le = pe[min(q + 1, m - 1) * I + i];
        const T a = q >= 1 ? la : (T)0;
        const T bq = (q >= 1 && q - 1 < nodd) ? lb : (T)0;
        const T c = lc;
        const T d = q < nodd ? ld : (T)0;
        const T e = q + 1 < m ? le : (T)0;
        const T w0 = mt[0 * m + q], w1 = mt[1 * m + q], w2 = mt[2 * m + q], w3 = mt[3 * m + q],
                w4 = mt[4 * m + q], w5 = mt[5 * m + q], w6 = mt[6 * m + q], r1w = mt[7 * m + q],
                r4w = mt[8 * m + q];
        const T tb = a * w0 + bq * w1 + c * w2;
        T tc = bq * w2 + c * w3 + d * w4;
        const T td = c * w4 + d * w5 + e * w6;
        tc += tb * r1w + td * r4w;
        po_out[r] = tc;
      }
      r += 256;
      if (I >= 256) {
        i += 256;
        if (i >= I) i -= I, q++;
      } else {
        q = r / I;
        i = r - q * I;
      }
    }
  }
}

// k_nd_lpk along the FASTEST dimension (the first sweep: strided input with the zero rule), lane =
// output index q. Rows in groups as in k_nd_coeff_rows (dimensions 0..2 once per group), the nine
// weights of a lane once per group (they depend on q alone), and of the five inputs of an output
// two loaded -- the coarse node q and the odd node behind it -- the other three the neighbouring
// lanes' (q - 1 of both, q + 1 of the coarse ones), zeros already in place of what the rules leave
// out. Two rows in flight.
struct NdFastSweep {
  uint32_t e[kNd - 1];    // extents of the slow dimensions, right-aligned
  uint64_t is[kNd - 1];   // their input strides
  uint32_t mc[kNd - 1];   // coarse extents (zero rule)
  uint64_t sf;            // input stride along the fastest dimension
  uint32_t n, m;          // fine / coarse size of the fastest dimension
  int zero_all_coarse;
  uint32_t gsz, gpl, groups;
};

template <typename T>
__global__ void __launch_bounds__(256)
k_nd_lpk_fast(NdFastSweep s, const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ mt) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + blockIdx.x * 4;
  const uint32_t nwave = gridDim.x * 4;
  const uint32_t m = s.m, nodd = s.n - s.m;
  const uint32_t olast = nodd ? nodd - 1 : 0;
  const int sh = m > 64 ? 1 : 0;  // longer than the wave: 62 outputs a round, lanes 0 and 63 partners only
  for (uint32_t g = wave; g < s.groups; g += nwave) {
    uint32_t line = g / s.gpl;
    const uint32_t q0 = (g - line * s.gpl) * s.gsz, q1 = min(s.e[3], q0 + s.gsz);
    const uint64_t row0 = (uint64_t)line * s.e[3];  // output row of dimension-3 position 0
    uint64_t base012 = 0;
    bool ez012 = s.zero_all_coarse;
#pragma unroll
    for (int d = 2; d >= 0; d--) {
      const uint32_t qq = line / s.e[d];
      const uint32_t pos = line - qq * s.e[d];
      line = qq;
      base012 += pos * s.is[d];
      if (pos >= s.mc[d]) ez012 = false;
    }
    for (uint32_t c0 = 0; c0 < m; c0 += sh ? 62 : 64) {
      const int qi = (int)c0 + lane - sh;
      const bool valid = qi >= 0 && (uint32_t)qi < m;
      const bool own = valid && (!sh || (lane >= 1 && lane <= 62));
      const uint32_t q = valid ? (uint32_t)qi : 0;
      T wk[9];
#pragma unroll
      for (int k = 0; k < 9; k++) wk[k] = mt[k * m + q];
      const uint64_t oc = (uint64_t)q * s.sf;                                   // the coarse node q ...
      const uint64_t oo = (nodd ? (uint64_t)(m + min(q, olast)) : (uint64_t)q) * s.sf;  // ... the odd one behind it
      const bool has_d = valid && q < nodd;
      auto row = [&](uint32_t q3) -> T {
        const T *pr = in + base012 + q3 * s.is[3];
        const bool ez = ez012 && q3 < s.mc[3];
        const T lc = pr[oc], ld = pr[oo];
        const T c = (valid && !ez) ? lc : (T)0;
        const T d = has_d ? ld : (T)0;
        const T a = lane_below(c), bq = lane_below(d), e = lane_above(c);  // (lane 0 / 63: own value back, unused or zero by construction below)
        const T a0 = (lane >= 1 && q >= 1) ? a : (T)0;
        const T b0 = (lane >= 1 && q >= 1) ? bq : (T)0;
        const T e0 = (lane <= 62 && q + 1 < m) ? e : (T)0;
        const T tb = a0 * wk[0] + b0 * wk[1] + c * wk[2];
        T tc = b0 * wk[2] + c * wk[3] + d * wk[4];
        const T td = c * wk[4] + d * wk[5] + e0 * wk[6];
        tc += tb * wk[7] + td * wk[8];
        return tc;
      };
      uint32_t q3 = q0;
      for (; q3 + 1 < q1; q3 += 2) {
        T *o0 = out + (row0 + q3) * m, *o1 = o0 + m;
        const T x0 = row(q3), x1 = row(q3 + 1);
        if (own) o0[q] = x0, o1[q] = x1;
      }
      if (q3 < q1) {
        T *o0 = out + (row0 + q3) * m;
        const T x0 = row(q3);
        if (own) o0[q] = x0;
      }
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
