// Decompression side of the fused path (dequantize + recompose), gfx950.
//
//   k_level_loadvec_q : load vector Lr(Lc(Lf(C))) of level l straight from the QUANTIZED
//                       coefficients (dequantized on the fly, never materialised as floats);
//                       same marching / sweep structure as k_level_fused (kernels_fused.hpp),
//                       minus the coefficient computation.
//   k_level_restore_q : fine nodal array from the corrected coarse nodes and the quantized
//                       coefficients: even nodes are copies, odd nodes = coefficient +
//                       interpolant (GpkRev3D, GridProcessingKernel3D.hpp:1231-2352), with the
//                       dequantizer of LinearQuantization.hpp:246-264 fused in.
//   k_head_in_q       : level-0 nodal values out of the head of the quantized array.
// Between the two, the Thomas solves of kernels_ipk.hpp subtract the correction from the
// coarse nodes. Bit-identical to dequantize + recompose with kernels_v1.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace mgh {

template <typename T> struct RecomposeArgs {
  int n[3], m[3];
  const int64_t *q;  // quantized coefficients, reordered layout, strides (dI, dJ, 1)
  const T *coef;     // ... or the coefficients themselves (kernels instantiated with QT = T)
  const uint16_t *q16;  // ... or 16-bit dictionary symbols (QT = uint16_t): symbol 0 may stand for
                        // an out-of-dictionary value, looked up by its linear index in the table
  const unsigned long long *oh_key;  // open-addressing table: key = linear index + 1, 0 = empty
  const long long *oh_val;           // shifted quantized value of the outlier
  uint32_t oh_mask;                  // slots - 1 (power of two)
  size_t dI, dJ;
  const T *coarse;   // compact (m0, m1, m2), corrected coarse nodes (restore only)
  T *load;           // compact (m0, m1, m2) (loadvec only)
  T *fine;           // natural fine box, strides (fI, fJ, 1) (restore only)
  size_t fI, fJ;
  const T *ratio[3];
  const T *mass[3];
  T qv;              // quantizer * reciprocal volume of this level (dequantize factor)
  int64_t half;      // dict_size / 2 if the Huffman shift was applied, else 0
  // D = 4 (recompose_levels4, capi.hip): the kernels work on ONE slice of the slowest dimension t
  size_t lin_base;   // element offset of the slice inside q / coef / q16
  int allcoef;       // odd t-slice: every node of the slice is a coefficient of this level
  const T *coarse_b; // odd t-slice: corrected coarse nodes of the slice above (coarse: below)
  const T *ratio_t;  // ... interpolation ratios along t of the level,
  int tpos;          // ... and the slice's (odd) position: ratio_t[tpos - 1] applies
  // ... or, with zb_mode != 0, on ALL slices of a kind in one launch (64^4: 131 + 131 launches of
  // 5-9 us each a level loop otherwise): the slice is the workgroup's, the fields above are set
  // from it by slice_batch() at the head of the kernel. 1: load vectors, slice = padded t position
  // P (blockIdx.z = P * zb_nz + r-chunk); 2 / 3: node restore of the even / odd slices (blockIdx.z).
  int zb_mode, zb_mt, zb_nt, zb_nz;
  size_t zb_sT;      // elements between t-slices of the coefficient source
  size_t zb_M;       // ... of the compact coarse / load arrays
  size_t zb_fT;      // ... of the fine array
};

// false: the workgroup's slice has no work (the ghost slice of an even extent: zeroed by the host)
template <typename T> __device__ __forceinline__ bool slice_batch(RecomposeArgs<T> &A, int *zchunk) {
  *zchunk = (int)blockIdx.z;
  if (A.zb_mode == 0) return true;
  if (A.zb_mode == 1) {
    const int P = (int)blockIdx.z / A.zb_nz;
    *zchunk = (int)blockIdx.z - P * A.zb_nz;
    if (A.zb_nt % 2 == 0 && P == A.zb_nt - 1) return false;
    A.allcoef = P & 1;
    A.lin_base = (size_t)((P & 1) ? A.zb_mt + (P - 1) / 2 : P / 2) * A.zb_sT;
    A.load += (size_t)P * A.zb_M;
    return true;
  }
  const int zi = (int)blockIdx.z;
  *zchunk = 0;
  if (A.zb_mode == 2) {
    const int tp = (A.zb_nt % 2 == 0 && zi == A.zb_mt - 1) ? A.zb_nt - 1 : 2 * zi;
    A.fine += (size_t)tp * A.zb_fT;
    A.coarse += (size_t)zi * A.zb_M;
    A.lin_base = (size_t)zi * A.zb_sT;
  } else {
    const int tp = 2 * zi + 1;
    A.fine += (size_t)tp * A.zb_fT;
    A.coarse += (size_t)zi * A.zb_M;
    A.coarse_b = A.coarse + A.zb_M;
    A.tpos = tp;
    A.lin_base = (size_t)(A.zb_mt + zi) * A.zb_sT;
  }
  return true;
}

template <typename T> __device__ __forceinline__ T dequant_one(int64_t qd, int64_t half, T qv) {
  const int64_t d = qd - half;
  // (T)d through the 32-bit converter when the whole wave's values fit (they practically
  // always do): same value, a fraction of the instructions of the 64-bit conversion
  if (__all(d == (int64_t)(int)d)) return qv * (T)(int)d;
  return qv * (T)d;
}

// The kernels read the level's coefficients either as quantized integers (QT = int64_t, the
// dequantizer is applied on the fly) or as floating-point coefficients (QT = T: Recompose on
// its own, Compressor::Recompose).
template <typename T> __device__ __forceinline__ const int64_t *qsrc(const RecomposeArgs<T> &A, int64_t) { return A.q; }
template <typename T> __device__ __forceinline__ const T *qsrc(const RecomposeArgs<T> &A, T) { return A.coef; }
template <typename T> __device__ __forceinline__ int64_t qmissing(const RecomposeArgs<T> &A, int64_t) { return A.half; }
template <typename T> __device__ __forceinline__ T qmissing(const RecomposeArgs<T> &, T) { return (T)0; }
template <typename T> __device__ __forceinline__ T qdecode(const RecomposeArgs<T> &A, int64_t v) {
  return dequant_one<T>(v, A.half, A.qv);
}
template <typename T> __device__ __forceinline__ T qdecode(const RecomposeArgs<T> &, T v) { return v; }

// 16-bit symbols travel as 32-bit values (kMissing16 where a window element has no
// coefficient) and are dequantized where they are used: d = symbol - half fits an int, same
// conversion and product as dequant_one. Symbol 0 is looked up in the outlier table by its
// linear index (rare: outliers and true zeros only). The loads themselves stay plain loads --
// a lookup inside the load loop would put every load into its own basic block and serialise them.
constexpr uint32_t kMissing16 = 0xffffffffu;
__device__ __forceinline__ uint32_t outlier_hash(unsigned long long lin) {
  return (uint32_t)((lin * 0x9E3779B97F4A7C15ull) >> 32);
}
template <typename T>
__device__ __forceinline__ T qdecode(const RecomposeArgs<T> &A, uint32_t v, size_t lin) {
  if (v == kMissing16) return (T)0;
  if (v == 0 && A.oh_key) {
    uint32_t slot = outlier_hash(lin) & A.oh_mask;
    for (uint32_t probes = 0; probes <= A.oh_mask; probes++) {
      const unsigned long long k = A.oh_key[slot];
      if (k == (unsigned long long)lin + 1) return A.qv * (T)((int64_t)A.oh_val[slot] - A.half);
      if (k == 0) break;
      slot = (slot + 1) & A.oh_mask;
    }
  }
  return A.qv * (T)((int)v - (int)A.half);
}
template <typename T> __device__ __forceinline__ T qdecode(const RecomposeArgs<T> &A, int64_t v, size_t) {
  return qdecode<T>(A, v);
}
__device__ __forceinline__ float qdecode(const RecomposeArgs<float> &, float v, size_t) { return v; }
__device__ __forceinline__ double qdecode(const RecomposeArgs<double> &, double v, size_t) { return v; }
template <typename T> __device__ __forceinline__ uint32_t qload(const RecomposeArgs<T> &, const uint16_t *p) { return *p; }
template <typename T> __device__ __forceinline__ int64_t qload(const RecomposeArgs<T> &, const int64_t *p) { return *p; }
template <typename T> __device__ __forceinline__ T qload(const RecomposeArgs<T> &, const T *p) { return *p; }
template <typename T> __device__ __forceinline__ const uint16_t *qsrc(const RecomposeArgs<T> &A, uint16_t) { return A.q16; }
template <typename T> __device__ __forceinline__ uint32_t qmissing(const RecomposeArgs<T> &, uint16_t) { return kMissing16; }
// type a loaded value travels in
template <typename T, typename QT> struct QReg { using type = QT; };
template <typename T> struct QReg<T, uint16_t> { using type = uint32_t; };

template <typename T, typename QT, int TC, int TF, int RCH>
__global__ void __launch_bounds__(256)
k_level_loadvec_q(RecomposeArgs<T> A) {
  constexpr int WC = 2 * TC + 3;
  constexpr int WF = 2 * TF + 3;
  constexpr int HF = TF + 2;
  constexpr int ROW = 2 * HF;
  constexpr int NT = 256;
  static_assert(TC * TF == NT, "one c-sweep output per thread");
  constexpr int NL = (WC * WF + NT - 1) / NT;
  __shared__ T Cs[WC * ROW];
  __shared__ T t1s[WC][TF + 1];
  __shared__ T wrs[RCH][9];
#define LI(lc, lf) ((lc) * ROW + ((lf) & 1) * HF + ((lf) >> 1))
  const int tid = threadIdx.x;
  int zchunk;
  if (!slice_batch(A, &zchunk)) return;
  const int F0 = blockIdx.x * TF, C0 = blockIdx.y * TC, R0 = zchunk * RCH;
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int c_lo = 2 * C0 - 2, f_lo = 2 * F0 - 2, r_lo = 2 * R0 - 2;
  const int r_hi = min(2 * R0 + 2 * RCH, 2 * mr);
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2, Pmax_f = 2 * mf - 2;
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int ghost_f = (nf % 2 == 0) ? nf - 1 : -7;
  for (int e = tid; e < RCH * 9; e += NT) {
    const int R = R0 + e / 9, k = e % 9;
    wrs[e / 9][k] = R < mr ? A.mass[0][k * mr + R] : (T)0;
  }
  const int jf = tid % TF, jc = tid / TF;
  T wf[9], wc[9];
  {
    const int Jf = F0 + jf, Jc = C0 + jc;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      wf[k] = Jf < mf ? A.mass[2][k * mf + Jf] : (T)0;
      wc[k] = Jc < mc ? A.mass[1][k * mc + Jc] : (T)0;
    }
  }
  T win[5] = {0, 0, 0, 0, 0};
  // window elements of this thread: offset of the coefficient inside an r-plane of q, parity
  constexpr uint32_t kNone = 0xffffffffu;
  uint32_t qoff[NL];
  int lds[NL];
  bool evn[NL];  // even in c and f: a coefficient only on odd r-planes
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const int e = tid + k * NT;
    const int lc = e / WF, lf = e - lc * WF;
    const int Pc = c_lo + lc, Pf = f_lo + lf;
    const bool ok = e < WC * WF && Pc >= 0 && Pc <= Pmax_c && Pf >= 0 && Pf <= Pmax_f &&
                    Pc != ghost_c && Pf != ghost_f;
    const int oj = (Pc & 1) ? mc + (Pc - 1) / 2 : Pc / 2;
    const int okk = (Pf & 1) ? mf + (Pf - 1) / 2 : Pf / 2;
    qoff[k] = ok ? (uint32_t)(oj * (int)A.dJ + okk) : kNone;
    lds[k] = e < WC * WF ? LI(lc, lf) : -1;
    evn[k] = !(lc & 1) && !(lf & 1);
  }
  using QR = typename QReg<T, QT>::type;
  auto fetch = [&](int p, QR(&reg)[NL]) {
    const bool pv = p >= 0 && p <= Pmax_r && p != ghost_r;
    const bool p_odd = p & 1;
    const int oi = p_odd ? mr + (p - 1) / 2 : p / 2;
    const QT *base = qsrc<T>(A, QT()) + A.lin_base + (size_t)(pv ? oi : 0) * A.dI;
#pragma unroll
    for (int k = 0; k < NL; k++)
      reg[k] = (pv && qoff[k] != kNone && (p_odd || !evn[k] || A.allcoef)) ? qload<T>(A, base + qoff[k]) : qmissing<T>(A, QT());
  };
  QR cur[NL], nxt[NL];
  fetch(r_lo, cur);
  for (int p = r_lo; p <= r_hi; p++) {
    if (p < r_hi) fetch(p + 1, nxt);
    // Phase A': dequantized coefficient field of the window (0 at coarse / missing nodes:
    // a missing value was fetched as `half`, which dequantizes to exactly 0)
    {
      const size_t pb = A.lin_base + (size_t)((p & 1) ? mr + (p - 1) / 2 : p / 2) * A.dI;  // (index of a looked-up value)
#pragma unroll
      for (int k = 0; k < NL; k++)
        if (lds[k] >= 0) Cs[lds[k]] = qdecode(A, cur[k], pb + qoff[k]);
    }
    __syncthreads();
    for (int lc = jc; lc < WC; lc += TC) {
      const T *row = Cs + lc * ROW;
      t1s[lc][jf] = mass_apply(row[jf], row[HF + jf], row[jf + 1], row[HF + jf + 1], row[jf + 2], wf);
    }
    __syncthreads();
    {
      const int lc = 2 * jc + 2;
      const T v = mass_apply(t1s[lc - 2][jf], t1s[lc - 1][jf], t1s[lc][jf], t1s[lc + 1][jf],
                             t1s[lc + 2][jf], wc);
      win[0] = win[1];
      win[1] = win[2];
      win[2] = win[3];
      win[3] = win[4];
      win[4] = v;
    }
    if (!(p & 1) && p >= 2 * R0 + 2) {
      const int R = (p - 2) / 2;
      const int Jc = C0 + jc, Jf = F0 + jf;
      if (R < mr && Jc < mc && Jf < mf) {
        T wr[9];
#pragma unroll
        for (int k = 0; k < 9; k++) wr[k] = wrs[R - R0][k];
        A.load[((size_t)R * mc + Jc) * mf + Jf] =
            mass_apply(win[0], win[1], win[2], win[3], win[4], wr);
      }
    }
#pragma unroll
    for (int k = 0; k < NL; k++) cur[k] = nxt[k];
  }
#undef LI
}

// One wave per fine row (rp, cp); a lane handles the node PAIRS (fp = 2t, 2t + 1) for
// t = lane, lane + 64, ...: consecutive lanes read consecutive quantized values in both the
// coarse-f and the coefficient-f part of the reordered row and consecutive coarse nodes, and
// write 8 contiguous bytes each. Row-level index math is done once per wave.
template <typename T, typename QT>
__global__ void __launch_bounds__(256)
k_level_restore_q(RecomposeArgs<T> A) {
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int cp = blockIdx.y * blockDim.y + threadIdx.y;
  const int rp = blockIdx.z;
  if (cp >= nc || rp >= nr) return;
  // coarse index of an even fine position (or of the real last node of an even-sized dim),
  // else the node is a coefficient node at odd position p
  auto split = [](int p, int n, int m, bool &odd) -> int {
    odd = (p & 1) && !(n % 2 == 0 && p == n - 1);
    return odd ? m + (p - 1) / 2 : (p == n - 1 ? m - 1 : p / 2);
  };
  bool ro, co;
  const int i = split(rp, nr, mr, ro), j = split(cp, nc, mc, co);
  const size_t mJ = mf, mI = (size_t)mc * mf;
  const int r0 = ro ? (rp - 1) / 2 : i, c0 = co ? (cp - 1) / 2 : j;
  const T rr = ro ? A.ratio[0][rp - 1] : (T)0, rc = co ? A.ratio[1][cp - 1] : (T)0;
  const size_t qlin = A.lin_base + (size_t)i * A.dI + (size_t)j * A.dJ;  // (lin_base: the t-slice of a 4-D level)
  const QT *qrow = qsrc<T>(A, QT()) + qlin;
  T *out = A.fine + (size_t)rp * A.fI + (size_t)cp * A.fJ;
  const bool pure_coarse = !ro && !co;
  const T *rows[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
      rows[a][b] = A.coarse + (size_t)(r0 + (ro ? a : 0)) * mI + (size_t)(c0 + (co ? b : 0)) * mJ;
  const bool pair_aligned = (reinterpret_cast<uintptr_t>(out) & (2 * sizeof(T) - 1)) == 0;
  const int npair = (nf + 1) / 2;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < npair; t += gridDim.x * blockDim.x) {
    // node E = fine 2t (coarse-f index t); node O = fine 2t+1: a coefficient node in f, or
    // -- last node of an even-sized dim -- the coarse-f node t+1 = mf-1
    const int fpO = 2 * t + 1;
    const bool hasO = fpO < nf;
    const bool fo = hasO && !(nf % 2 == 0 && fpO == nf - 1);
    const T rf = fo ? A.ratio[2][fpO - 1] : (T)0;
    const int t1 = min(t + 1, mf - 1);
    // interpolants of both nodes from the (up to) 4 coarse rows, f innermost, then c, then r
    T hE[2], hO[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      if (a == 1 && !ro) break;
      T gE[2], gO[2];
#pragma unroll
      for (int b = 0; b < 2; b++) {
        if (b == 1 && !co) break;
        const T v0 = rows[a][b][t], v1 = rows[a][b][t1];
        gE[b] = v0;
        gO[b] = fo ? lerp_ref(v0, v1, rf) : v1;
      }
      hE[a] = co ? lerp_ref(gE[0], gE[1], rc) : gE[0];
      hO[a] = co ? lerp_ref(gO[0], gO[1], rc) : gO[0];
    }
    const T iE = ro ? lerp_ref(hE[0], hE[1], rr) : hE[0];
    const T iO = ro ? lerp_ref(hO[0], hO[1], rr) : hO[0];
    // E: coarse node (pure copy) unless r or c is odd
    T vE = iE;
    if (!pure_coarse) vE = qdecode(A, qload<T>(A, qrow + t), qlin + t) + iE;
    T vO = iO;  // (pure coarse last node of an even-sized dim: iO = row[mf-1])
    if (hasO) {
      if (fo)
        vO = qdecode(A, qload<T>(A, qrow + mf + t), qlin + mf + t) + iO;
      else if (!pure_coarse)
        vO = qdecode(A, qload<T>(A, qrow + mf - 1), qlin + mf - 1) + iO;
    }
    // the pair is contiguous: one 2-element store when the row start allows it
    T *dst = out + 2 * t;
    if (hasO && pair_aligned) {
      struct alignas(2 * sizeof(T)) Pair { T a, b; };
      *reinterpret_cast<Pair *>(dst) = Pair{vE, vO};
    } else {
      dst[0] = vE;
      if (hasO) dst[1] = vO;
    }
  }
}

// The same for a PAIR of fine rows (cp = 2J, 2J + 1) per wave: both rows interpolate from the
// coarse rows J (and J + 1), so the coarse values and their f-interpolants are loaded / formed
// once for the two, and the row-level set-up is paid once (466 instead of 565 us over the levels
// of 512^3). Every output is computed with the operations of k_level_restore_q, in the same
// order. (A 2 x 2 group -- two planes x two rows per wave -- was slower again: 570 us.)
//
// TODD (D = 4, an ODD slice of the slowest dimension t): every node of the slice is a coefficient
// node of the level -- value = coefficient + lerp_t(X_a, X_b), X = the 3-D interpolant (f, then c,
// then r) of the coarse slice below (A.coarse) / above (A.coarse_b), the mirror of the TODD tiles
// of kernels_fused2.hpp (CalcCoefficientsND.hpp:25-236: nested lerps, fastest dim innermost).
template <typename T, typename QT, bool TODD = false>
__global__ void __launch_bounds__(256)
k_level_restore2_q(RecomposeArgs<T> A) {
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int J = blockIdx.y * blockDim.y + threadIdx.y;  // pair index: rows 2J and 2J + 1
  const int rp = blockIdx.z;
  const int cpE = 2 * J, cpO = 2 * J + 1;
  if (cpE >= nc || rp >= nr) return;
  const bool hasRowO = cpO < nc;
  const bool ro = (rp & 1) && !(nr % 2 == 0 && rp == nr - 1);
  const int i = ro ? mr + (rp - 1) / 2 : (rp == nr - 1 ? mr - 1 : rp / 2);
  const int r0 = ro ? (rp - 1) / 2 : i;
  // row E is a coarse row in c (even position); row O is a coefficient row unless it is the real
  // last node of an even-sized dim (then it is the coarse row mc - 1 = J + 1)
  const bool coO = hasRowO && !(nc % 2 == 0 && cpO == nc - 1);
  const int jE = J;
  const int jO = coO ? mc + J : mc - 1;
  const T rr = ro ? A.ratio[0][rp - 1] : (T)0, rc = coO ? A.ratio[1][cpO - 1] : (T)0;
  const size_t mJ = mf, mI = (size_t)mc * mf;
  const size_t qlinE = A.lin_base + (size_t)i * A.dI + (size_t)jE * A.dJ,
               qlinO = A.lin_base + (size_t)i * A.dI + (size_t)jO * A.dJ;
  const QT *qrowE = qsrc<T>(A, QT()) + qlinE, *qrowO = qsrc<T>(A, QT()) + qlinO;
  T *outE = A.fine + (size_t)rp * A.fI + (size_t)cpE * A.fJ;
  T *outO = outE + A.fJ;
  const bool pureE = !TODD && !ro;          // row E: coarse in c; a pure copy unless r is odd
  const bool pureO = !TODD && !ro && !coO;  // (row O as the coarse last row of an even-sized dim)
  const int cJ1 = min(J + 1, mc - 1);
  const T *rowJ[2], *rowJ1[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    rowJ[a] = A.coarse + (size_t)(r0 + (ro ? a : 0)) * mI + (size_t)J * mJ;
    rowJ1[a] = A.coarse + (size_t)(r0 + (ro ? a : 0)) * mI + (size_t)cJ1 * mJ;
  }
  const ptrdiff_t to_b = TODD ? A.coarse_b - A.coarse : 0;
  const T rt = TODD ? A.ratio_t[A.tpos - 1] : (T)0;
  const bool alE = (reinterpret_cast<uintptr_t>(outE) & (2 * sizeof(T) - 1)) == 0;
  const bool alO = (reinterpret_cast<uintptr_t>(outO) & (2 * sizeof(T) - 1)) == 0;
  const int npair = (nf + 1) / 2;
  struct alignas(2 * sizeof(T)) Pair { T a, b; };
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < npair; t += gridDim.x * blockDim.x) {
    const int fpO = 2 * t + 1;
    const bool hasO = fpO < nf;
    const bool fo = hasO && !(nf % 2 == 0 && fpO == nf - 1);
    const T rf = fo ? A.ratio[2][fpO - 1] : (T)0;
    const int t1 = min(t + 1, mf - 1);
    // the four interpolants (row E: nodes E, O; row O: nodes E, O) from one coarse slice
    // (`off` = element offset of the slice relative to A.coarse)
    auto interp = [&](ptrdiff_t off, T &iEE, T &iEO, T &iOE, T &iOO) {
      // f-level values of the coarse rows J and J+1 at the (up to) two r-planes
      T eJ[2], oJ[2], eJ1[2], oJ1[2];
#pragma unroll
      for (int a = 0; a < 2; a++) {
        if (a == 1 && !ro) break;
        const T v0 = rowJ[a][off + t], v1 = rowJ[a][off + t1];
        eJ[a] = v0;
        oJ[a] = fo ? lerp_ref(v0, v1, rf) : v1;
        if (hasRowO) {
          const T w0 = rowJ1[a][off + t], w1 = rowJ1[a][off + t1];
          eJ1[a] = w0;
          oJ1[a] = fo ? lerp_ref(w0, w1, rf) : w1;
        }
      }
      // row E (c even): the row-J values, then r
      iEE = ro ? lerp_ref(eJ[0], eJ[1], rr) : eJ[0];
      iEO = ro ? lerp_ref(oJ[0], oJ[1], rr) : oJ[0];
      // row O: c-interpolation between rows J and J+1 (or the coarse last row J+1), then r
      if (hasRowO) {
        T hE[2], hO[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
          if (a == 1 && !ro) break;
          hE[a] = coO ? lerp_ref(eJ[a], eJ1[a], rc) : eJ1[a];
          hO[a] = coO ? lerp_ref(oJ[a], oJ1[a], rc) : oJ1[a];
        }
        iOE = ro ? lerp_ref(hE[0], hE[1], rr) : hE[0];
        iOO = ro ? lerp_ref(hO[0], hO[1], rr) : hO[0];
      }
    };
    T iEE, iEO, iOE = 0, iOO = 0;
    interp(0, iEE, iEO, iOE, iOO);
    if (TODD) {
      T bEE, bEO, bOE = 0, bOO = 0;
      interp(to_b, bEE, bEO, bOE, bOO);
      iEE = lerp_ref(iEE, bEE, rt);
      iEO = lerp_ref(iEO, bEO, rt);
      iOE = lerp_ref(iOE, bOE, rt);
      iOO = lerp_ref(iOO, bOO, rt);
    }
    // ---- row E
    {
      T vE = iEE;
      if (!pureE) vE = qdecode(A, qload<T>(A, qrowE + t), qlinE + t) + iEE;
      T vO = iEO;
      if (hasO) {
        if (fo) vO = qdecode(A, qload<T>(A, qrowE + mf + t), qlinE + mf + t) + iEO;
        else if (!pureE) vO = qdecode(A, qload<T>(A, qrowE + mf - 1), qlinE + mf - 1) + iEO;
      }
      T *dst = outE + 2 * t;
      if (hasO && alE) {
        *reinterpret_cast<Pair *>(dst) = Pair{vE, vO};
      } else {
        dst[0] = vE;
        if (hasO) dst[1] = vO;
      }
    }
    // ---- row O
    if (hasRowO) {
      T vE = iOE;
      if (!pureO) vE = qdecode(A, qload<T>(A, qrowO + t), qlinO + t) + iOE;
      T vO = iOO;
      if (hasO) {
        if (fo) vO = qdecode(A, qload<T>(A, qrowO + mf + t), qlinO + mf + t) + iOO;
        else if (!pureO) vO = qdecode(A, qload<T>(A, qrowO + mf - 1), qlinO + mf - 1) + iOO;
      }
      T *dst = outO + 2 * t;
      if (hasO && alO) {
        *reinterpret_cast<Pair *>(dst) = Pair{vE, vO};
      } else {
        dst[0] = vE;
        if (hasO) dst[1] = vO;
      }
    }
  }
}

// D = 4: level-0 nodal values (compact (m0, m1, m2, m3)) out of the head of the quantized array;
// dT = element stride of t in that array.
template <typename T, typename QT>
__global__ void __launch_bounds__(256)
k_head_in4_q(int m0, int m1, int m2, int m3, RecomposeArgs<T> A, size_t dT, T *__restrict__ nodal) {
  const int total = m0 * m1 * m2 * m3;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int k = e % m3, j = (e / m3) % m2, i = (e / (m3 * m2)) % m1, t = e / (m3 * m2 * m1);
    const size_t lin = (size_t)t * dT + (size_t)i * A.dI + (size_t)j * A.dJ + k;
    nodal[e] = qdecode(A, qload<T>(A, qsrc<T>(A, QT()) + lin), lin);
  }
}

template <typename T, typename QT>
__global__ void __launch_bounds__(256)
k_head_in_q(int m0, int m1, int m2, RecomposeArgs<T> A, T *__restrict__ nodal) {
  const int total = m0 * m1 * m2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
    const size_t lin = (size_t)i * A.dI + (size_t)j * A.dJ + k;
    nodal[e] = qdecode(A, qload<T>(A, qsrc<T>(A, QT()) + lin), lin);
  }
}

// Outlier table for the 16-bit symbol path: idx -> value, open addressing with linear probing;
// indices outside the array (damaged streams) are dropped. keys[] zero before the launch.
__global__ void __launch_bounds__(256)
k_outlier_hash_build(const uint64_t *__restrict__ oidx, const int64_t *__restrict__ oval, size_t count,
                     size_t total, unsigned long long *__restrict__ keys, long long *__restrict__ vals,
                     uint32_t mask) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const unsigned long long lin = oidx[i];
  if (lin >= total) return;
  uint32_t slot = outlier_hash(lin) & mask;
  for (uint32_t probes = 0; probes <= mask; probes++) {
    const unsigned long long prev = atomicCAS(&keys[slot], 0ull, lin + 1);
    if (prev == 0 || prev == lin + 1) {
      vals[slot] = oval[i];
      return;
    }
    slot = (slot + 1) & mask;
  }
}

} // namespace mgh
