// What the fused level kernels share (kernels_fused2.hpp: the marching tile kernel,
// kernels_box.hpp: the box kernel, kernels_tail.hpp): the argument block of one level pass, the
// mass / restriction stencil with host-prepared constants, the quantizer with wave-aggregated
// outlier slots, the head of the output and the device-side quantizer table.
// One pass over the level-l nodal array produces
//   (a) the coarse nodal values (level l-1, before correction),
//   (b) the level-l multilevel coefficients, either as T in the reordered
//       layout or already quantized to int64 (+ outliers),
//   (c) the load vector  Lr(Lc(Lf(C)))  of the correction, i.e. the output of
//       the three fused mass-matrix/restriction sweeps,
// replacing CopyND + GpkReo3D + Lpk1/2/3Reo3D (+ the quantizer for this
// level's coefficients) of the reference
// (DataRefactoring.hpp:80-109, GridProcessingKernel3D.hpp:21-1179,
//  LinearProcessingKernel3D.hpp:27-1048, LinearQuantization.hpp:146-245).
// Everything is expressed in PADDED fine coordinates P in [0, 2m-2] per dim
// (m = coarse size): for an even-sized dim the real last node sits at P = n and
// P = n-1 is the ghost node whose coefficient is zero (Hierarchy.hpp:38-42,
// LinearProcessingKernel3D.hpp:52,177-203). All arithmetic keeps the reference's operation
// order (no FMA contraction): results are bit-identical to kernels_v1.hpp.
// (The first-generation marching kernel that used to live here was retired in round 5; the
// one-thread-per-element kernels of kernels_v1.hpp are the in-library cross-check.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_v1.hpp"

namespace mgh {

template <typename T> struct QParamArgs;  // (below)

template <typename T> struct FusedArgs {
  // level geometry: fine sizes n, coarse sizes m (r, c, f)
  int n[3], m[3];
  const T *u;      // fine nodal array, element strides (uI, uJ, 1)
  size_t uI, uJ;
  T *coarse;       // compact (m0, m1, m2)
  T *load;         // compact (m0, m1, m2): Lr(Lc(Lf(C)))
  // coefficient output (reordered layout, strides dI, dJ, 1)
  T *coef;         // OUT_T
  int64_t *q;      // OUT_Q
  uint16_t *q16;   // OUT_Q, instead of q: the dictionary symbols as 16-bit values (prep_huffman,
                   // dict_size <= 65536; out-of-dictionary values go to the outlier list only)
  size_t dI, dJ;
  const T *ratio[3];  // fine-level interpolation ratios
  const T *mass[3];   // mass_table SoA [9][m]
  // quantizer of this level (LinearQuantization.hpp:196-245): by value, or -- when the norm
  // never left the device -- read from qp[level] / qp[nlev + level] (k_make_qparams)
  T quantizer, volume;
  const T *qp;
  // ... or (round 6, MGH_INLINE_QP: the finest level of a REL call whose norm was just reduced)
  // computed by every workgroup itself from the reduction scalar `qslot` and the call's constants
  // `qinl` in device memory -- the k_make_qparams launch between the norm pass and this kernel
  // is gone; workgroup 0 leaves the table for the levels below
  const QParamArgs<T> *qinl;
  const unsigned long long *qslot;
  int level, nlev;
  int64_t dict_size;
  int prep_huffman;
  unsigned long long *outlier_count;
  uint64_t *outlier_idx;
  int64_t *outlier_val;
  unsigned long long outlier_cap;
  // OUT_NONE only: abs-max of the level's input (bit pattern, atomicMax), or nullptr
  unsigned long long *absmax_bits;
};

// OUT_NONE: coarse nodes + load vector only; the coefficients of the level are produced by
// k_level_emit (kernels_emit.hpp) on another stream
// OUT_QH (second-generation kernel only): OUT_Q with the options fixed at compile time to what
// the hot path runs -- dictionary shift on, int64 output -- so that the other variants' code and
// their scalar registers are not in the loop.
enum { OUT_T = 0, OUT_Q = 1, OUT_NONE = 2, OUT_QH = 3 };

template <typename T>
__device__ __forceinline__ T mass_apply(T a, T b, T c, T d, T e, const T (&w)[9]) {
  // Correction/LPKFunctor.h:77-93 with host-prepared constants (hierarchy.hpp)
  const T tb = a * w[0] + b * w[1] + c * w[2];
  T tc = b * w[2] + c * w[3] + d * w[4];
  const T td = c * w[4] + d * w[5] + e * w[6];
  tc += tb * w[7] + td * w[8];
  return tc;
}

// q = (int64) copysign(0.5 + |t * quantizer * volume|, t)  (LinearQuantization.hpp:203-207).
// Same value as quantize_one(); the float -> int64 conversion goes through the 32-bit
// converter whenever the magnitude allows it (it practically always does).
template <typename T>
__device__ __forceinline__ int64_t quantize_fast(T t, T quantizer, T volume) {
  const T a = (T)0.5 + abs_t(t * quantizer * volume);
  if (a < (T)2147483520.0) {
    const int r = (int)a;  // trunc, a >= 0.5
    return (int64_t)(t < 0 || (t == 0 && copysign_t((T)1, t) < 0) ? -r : r);
  }
  return (int64_t)copysign_t(a, t);
}

// Store up to NV quantized values of this lane; out-of-dictionary values of the whole wave
// get their outlier slots from ONE atomicAdd (LinearQuantization.hpp:208-241).
template <typename T, int NV>
__device__ __forceinline__ void emit_quantized(const FusedArgs<T> &A, const T (&v)[NV],
                                               const size_t (&lin)[NV], const bool (&on)[NV]) {
  int64_t qd[NV];
  bool outl[NV];
  bool any = false;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    qd[k] = quantize_fast(v[k], A.quantizer, A.volume);
    outl[k] = false;
    if (A.prep_huffman) {
      qd[k] += A.dict_size / 2;
      outl[k] = on[k] && !(qd[k] >= 0 && qd[k] < A.dict_size);
      any |= outl[k];
    }
  }
  if (__any(any)) {
    unsigned long long masks[NV];
    unsigned total = 0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      masks[k] = __ballot(outl[k]);
      total += __popcll(masks[k]);
    }
    unsigned long long base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == 0) base = atomicAdd(A.outlier_count, (unsigned long long)total);
    base = __shfl(base, 0, 64);
    unsigned before = 0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      if (outl[k]) {
        const unsigned rank = __popcll(masks[k] & ((1ULL << lane) - 1ULL));
        const unsigned long long o = base + before + rank;
        if (o < A.outlier_cap) {
          A.outlier_idx[o] = lin[k];
          A.outlier_val[o] = qd[k];
        }
        qd[k] = 0;
      }
      before += __popcll(masks[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NV; k++)
    if (on[k]) {
      if (A.q16) A.q16[lin[k]] = (uint16_t)qd[k];
      else A.q[lin[k]] = qd[k];
    }
}

// Quantize (or copy) the level-0 nodal values into the head of the output.
template <typename T, int OUT>
__global__ void __launch_bounds__(256)
k_head_out(int m0, int m1, int m2, const T *__restrict__ nodal, FusedArgs<T> A) {
  if (OUT == OUT_Q && A.qp) {
    A.quantizer = A.qp[0];
    A.volume = A.qp[A.nlev];
  }
  const int total = m0 * m1 * m2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int k = e % m2, j = (e / m2) % m1, i = e / (m2 * m1);
    const size_t lin = (size_t)i * A.dI + (size_t)j * A.dJ + k;
    const T v = nodal[e];
    if (OUT == OUT_T) {
      A.coef[lin] = v;
    } else {
      int64_t qd = quantize_one(v, A.quantizer, A.volume);
      if (A.prep_huffman) {
        qd += A.dict_size / 2;
        if (!(qd >= 0 && qd < A.dict_size)) {
          const unsigned long long o = atomicAdd(A.outlier_count, 1ULL);
          if (o < A.outlier_cap) {
            A.outlier_idx[o] = lin;
            A.outlier_val[o] = qd;
          }
          qd = 0;
        }
      }
      if (A.q16) A.q16[lin] = (uint16_t)qd;
      else A.q[lin] = qd;
    }
  }
}

// Quantizer table on the device (LinearQuantization.hpp:495-545) from a norm that stays on the
// device: qp[l] = 1 / (T)(abs_tol / den[l]), qp[nlev + l] = vol[l].
// abs_tol = 2 * tol * norm (REL), 2 * tol (ABS), or 2 * (T)(tol * norm) resp.
// 2 * sqrt((tol*norm)^2 / nsub) (decomposed domain, ErrorToleranceCalculator.hpp:134-155).
// All operations are single IEEE operations in the reference's order, so the values equal the
// host computation bit for bit. norm source: d_norm (T) if given, else the reduction scalar
// (absmax bits, or the double sum of squares).
constexpr int kMaxLevels = 40;
template <typename T> struct QParamArgs {
  const T *d_norm;                  // optional
  const unsigned long long *scalar;  // reduction result
  int s_is_inf, rel, decomposed, normalize;
  unsigned long long total, nsub;
  T tol;
  int nlev;
  double den[kMaxLevels];
  T vol[kMaxLevels];
  T *qp;
  T *norm_out;
  // launches saved: the outlier counter of this call and the norm scalar of the NEXT call
  // (the two scalar slots alternate) are zeroed here instead of by memsets
  unsigned long long *reset_count;
  unsigned long long *zero_next;
};

template <typename T>
__device__ __forceinline__ T qparams_norm(const QParamArgs<T> &P, const unsigned long long *scalar) {
  T norm;
  if (P.d_norm) {
    norm = *P.d_norm;
  } else if (P.s_is_inf) {
    const unsigned long long bits = *scalar;
    if (sizeof(T) == 4) norm = (T)__uint_as_float((unsigned)bits); else norm = (T)__longlong_as_double((long long)bits);
  } else {
    const double sum = __longlong_as_double((long long)*scalar);
    norm = (T)sum;
    if (sizeof(T) == 4) norm = P.normalize ? (T)sqrtf((float)(norm / (T)P.total)) : (T)sqrtf((float)norm);
    else norm = P.normalize ? (T)sqrt((double)(norm / (T)P.total)) : (T)sqrt((double)norm);
  }
  if (!P.d_norm && norm == 0) norm = sizeof(T) == 4 ? (T)1.1920928955078125e-7f : (T)2.220446049250313e-16;
  return norm;
}
// 2 x the absolute tolerance of the call (the numerator of every level's quantizer)
template <typename T> __device__ __forceinline__ double qparams_abs_tol2(const QParamArgs<T> &P, T norm) {
  double abs_tol;
  if (P.decomposed) {
    T lt;
    if (P.s_is_inf) {
      lt = P.rel ? P.tol * norm : P.tol;
    } else {
      const T a = P.rel ? (P.tol * norm) * (P.tol * norm) / (T)P.nsub : (P.tol * P.tol) / (T)P.nsub;
      lt = sizeof(T) == 4 ? (T)sqrtf((float)a) : (T)sqrt((double)a);
    }
    abs_tol = lt;
  } else {
    abs_tol = P.tol;
    if (P.rel) abs_tol *= norm;
  }
  abs_tol *= 2;
  return abs_tol;
}
template <typename T> __device__ __forceinline__ T qparams_level(double abs_tol2, double den) {
  T q = (T)(abs_tol2 / den);
  q = 1.0f / q;
  return q;
}

template <typename T>
__device__ __forceinline__ void make_qparams_body(const QParamArgs<T> &P) {
  const T norm = qparams_norm<T>(P, P.scalar);
  *P.norm_out = norm;
  const double abs_tol = qparams_abs_tol2<T>(P, norm);
  for (int l = 0; l < P.nlev; l++) {
    P.qp[l] = qparams_level<T>(abs_tol, P.den[l]);
    P.qp[P.nlev + l] = P.vol[l];
  }
  if (P.reset_count) *P.reset_count = 0;
  if (P.zero_next) *P.zero_next = 0;
}

// The finest level's own quantizer inside the level kernel (FusedArgs::qinl): every workgroup the
// same single IEEE operations on the same inputs as make_qparams_body -- the same bits; the first
// thread of the launch also leaves the table and the norm for the kernels behind it.
template <typename T>
__device__ __forceinline__ void inline_qparams(FusedArgs<T> &A, bool first_thread) {
  const QParamArgs<T> &P = *A.qinl;
  const T norm = qparams_norm<T>(P, A.qslot);
  const double abs_tol = qparams_abs_tol2<T>(P, norm);
  A.quantizer = qparams_level<T>(abs_tol, P.den[A.level]);
  A.volume = P.vol[A.level];
  if (first_thread) {
    *P.norm_out = norm;
    for (int l = 0; l < P.nlev; l++) {
      P.qp[l] = qparams_level<T>(abs_tol, P.den[l]);
      P.qp[P.nlev + l] = P.vol[l];
    }
  }
}

template <typename T> __global__ void k_make_qparams(QParamArgs<T> P) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  make_qparams_body<T>(P);
}

} // namespace mgh
