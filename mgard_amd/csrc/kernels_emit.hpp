// Coefficient + quantize pass of one level, without the correction sweeps ("emit pass").
//
// The correction chain of a level (load vector -> three Thomas solves -> next level) is a
// dependent sequence of latency-bound launches that want LDS, while computing and quantizing
// the level's coefficients is bandwidth-bound and nothing downstream waits for it. The driver
// therefore splits a big level in two: k_level_fused<OUT_NONE> feeds the chain (coarse nodes +
// load vector, and the abs-max of the input on the way), and the kernels below -- on a second
// stream, concurrently with the chain -- produce the quantized coefficients
// (GpkReo3D + LevelwiseLinearQuantizer of the reference: GridProcessingKernel3D.hpp:21-1179,
// LinearQuantization.hpp:146-245). Same cell arithmetic as kernels_fused.hpp phase A
// (interpolation order f, c, r; padded coordinates; ghost nodes), so the integers are
// identical to the unsplit path.
//
// To live beside the chain the pass uses NO LDS and no barriers: a wavefront owns one coarse
// plane R (fine planes 2R, 2R+1 and the even neighbour 2R+2), a strip of 64 cells along f
// (lane = cell; the 7 outputs of a cell row are 512-byte contiguous segments) and marches
// over a chunk of cells along c, carrying the shared even row and its f-interpolants in
// registers. The loads of the next DEPTH cells are in flight while a cell is computed, and
// the launch is persistent with a bounded number of wavefronts per CU, so that registers,
// LDS and wave slots stay free for the chain's workgroups. The 4 waves of a workgroup take 4
// consecutive R, so most of the neighbour-plane reads hit in the CU's / XCD's caches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace mgh {

// sign mask of x (all ones for negative values and -0)
__device__ __forceinline__ int sign_mask(float x) { return __float_as_int(x) >> 31; }
__device__ __forceinline__ int sign_mask(double x) { return __double2hiint(x) >> 31; }

// Store NV values of this lane at element index ubase[k] + lane_idx: ubase is wave-uniform
// (lives in SGPRs, so the address of every store is one scalar base + one 32-bit lane
// offset), the quantization goes through 32-bit integers and the common case -- every
// magnitude of the wave below 2^31 - 2^24, dictionary below 2^24 -- has no divergent branch.
// Values: q = (int64) copysign(0.5 + |t * quantizer * volume|, t), + dict/2, outliers outside
// [0, dict) (LinearQuantization.hpp:203-241), identical to emit_quantized().
template <typename T, int OUT, int NV>
__device__ __forceinline__ void emit_uniform(const FusedArgs<T> &A, const T (&v)[NV],
                                             const uint64_t (&ubase)[NV], uint32_t lane_idx,
                                             const bool (&on)[NV]) {
  if (OUT == OUT_T) {
#pragma unroll
    for (int k = 0; k < NV; k++)
      if (on[k]) (A.coef + ubase[k])[lane_idx] = v[k];
    return;
  }
  int q32[NV];
  bool big = false;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const T a = (T)0.5 + abs_t(v[k] * A.quantizer * A.volume);
    big |= !(a < (T)2130706432.0);  // 2^31 - 2^24; NaN lands here too
    const int r = (int)a, sm = sign_mask(v[k]);
    q32[k] = (r ^ sm) - sm;
  }
  if (__any(big) || A.dict_size > (1 << 24)) {  // rare: the general 64-bit path
    size_t lin[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) lin[k] = (size_t)ubase[k] + lane_idx;
    emit_quantized<T, NV>(A, v, lin, on);
    return;
  }
  if (A.prep_huffman) {
    const int half = (int)(A.dict_size / 2);
    const unsigned dict = (unsigned)A.dict_size;
    bool outl[NV];
    bool any = false;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      q32[k] += half;
      outl[k] = on[k] && (unsigned)q32[k] >= dict;
      any |= outl[k];
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
            A.outlier_idx[o] = ubase[k] + lane_idx;
            A.outlier_val[o] = (int64_t)q32[k];
          }
          q32[k] = 0;
        }
        before += __popcll(masks[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; k++)
    if (on[k]) (A.q + ubase[k])[lane_idx] = (int64_t)q32[k];
}

// load through a wave-uniform base and a 32-bit per-lane byte offset
template <typename T> __device__ __forceinline__ T ldu(const T *ubase, uint32_t byte_off) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + byte_off);
}

// Cells jf in [0, mf-1) (those with an odd-f member); work item = (R group of 4, strip, c chunk).
// Requires plane bytes < 4 GiB (32-bit lane offsets).
template <typename T, int OUT, int DEPTH>
__global__ void __launch_bounds__(256)
k_level_emit(FusedArgs<T> A, unsigned nstrip, unsigned ncch, unsigned cch, unsigned nwork) {
  if (OUT == OUT_Q && A.qp) {
    A.quantizer = A.qp[A.level];
    A.volume = A.qp[A.nlev + A.level];
  }
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> SGPR
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2;
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int ghost_f = (nf % 2 == 0) ? nf - 1 : -7;

  for (unsigned w = blockIdx.x; w < nwork; w += gridDim.x) {
    const unsigned strip = w % nstrip, chunk = (w / nstrip) % ncch, rg = w / (nstrip * ncch);
    const int R = (int)rg * 4 + wv;
    if (R >= mr) continue;  // (whole wave)
    const int jf = (int)strip * 64 + lane;
    const bool fval = jf < mf - 1;
    const int Pf0 = 2 * min(jf, mf - 1);
    const bool vf0 = fval, vf1 = fval && Pf0 + 1 != ghost_f;
    // byte offsets of the padded positions Pf0, Pf0 + 1, Pf0 + 2 inside a row
    const uint32_t o0 = (uint32_t)min(Pf0, nf - 1) * (uint32_t)sizeof(T),
                   o1 = (uint32_t)min(Pf0 + 1, nf - 1) * (uint32_t)sizeof(T),
                   o2 = (uint32_t)min(Pf0 + 2, nf - 1) * (uint32_t)sizeof(T);
    const T rf = Pf0 < nf ? A.ratio[2][Pf0] : (T)0;
    const T rr = 2 * R < nr ? A.ratio[0][2 * R] : (T)0;
    const bool odd_ok = 2 * R + 1 <= Pmax_r && 2 * R + 1 != ghost_r;
    const T *pe0 = A.u + (size_t)min(2 * R, nr - 1) * A.uI;
    const T *po = A.u + (size_t)min(2 * R + 1, nr - 1) * A.uI;
    const T *pe2 = A.u + (size_t)min(2 * R + 2, nr - 1) * A.uI;
    const int C0 = (int)(chunk * cch), C1 = min(C0 + (int)cch, mc);
    const uint64_t ob_e = (uint64_t)R * A.dI, ob_o = (uint64_t)(mr + R) * A.dI;

    // the 11 values a cell adds: E0 rows Pc0+1 (b0 b1), Pc0+2 (c0 c1 c2); O rows Pc0 (o00 o01),
    // Pc0+1 (o10 o11); E2 row Pc0+2 (e20 e22). Row bases are scalar, lane offsets 32-bit.
    auto load_cell = [&](int jc, T(&q)[11]) {
      const int Pc0 = 2 * min(jc, mc - 1);
      const size_t r0 = (size_t)min(Pc0, nc - 1) * A.uJ, r1 = (size_t)min(Pc0 + 1, nc - 1) * A.uJ,
                   r2 = (size_t)min(Pc0 + 2, nc - 1) * A.uJ;
      q[0] = ldu(pe0 + r1, o0);
      q[1] = ldu(pe0 + r1, o1);
      q[2] = ldu(pe0 + r2, o0);
      q[3] = ldu(pe0 + r2, o1);
      q[4] = ldu(pe0 + r2, o2);
      q[5] = ldu(po + r0, o0);
      q[6] = ldu(po + r0, o1);
      q[7] = ldu(po + r1, o0);
      q[8] = ldu(po + r1, o1);
      q[9] = ldu(pe2 + r2, o0);
      q[10] = ldu(pe2 + r2, o2);
    };
    T pre[DEPTH][11];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load_cell(C0 + d, pre[d]);
    // shared even row Pc0 = 2 C0 of both even planes, and its f-interpolants
    T a00, a01, a02, e00, e02;
    {
      const size_t r0 = (size_t)min(2 * C0, nc - 1) * A.uJ;
      a00 = ldu(pe0 + r0, o0);
      a01 = ldu(pe0 + r0, o1);
      a02 = ldu(pe0 + r0, o2);
      e00 = ldu(pe2 + r0, o0);
      e02 = ldu(pe2 + r0, o2);
    }
    T f0 = lerp_ref(a00, a02, rf), g0 = lerp_ref(e00, e02, rf);

    for (int jc = C0; jc < C1; jc++) {
      T q[11];
#pragma unroll
      for (int k = 0; k < 11; k++) q[k] = pre[0][k];
#pragma unroll
      for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
        for (int k = 0; k < 11; k++) pre[d][k] = pre[d + 1][k];
      load_cell(jc + DEPTH, pre[DEPTH - 1]);

      const int Pc0 = 2 * jc;
      const T rc = Pc0 < nc ? A.ratio[1][Pc0] : (T)0;
      const bool vc1 = Pc0 + 1 <= Pmax_c && Pc0 + 1 != ghost_c;
      // interpolation: f innermost, then c, then r (GridProcessingKernel3D.hpp:614-617,
      // 737-744, 854-871)
      const T f2 = lerp_ref(q[2], q[4], rf), g2 = lerp_ref(q[9], q[10], rf);
      const T x2 = lerp_ref(a00, q[2], rc), x3 = lerp_ref(f0, f2, rc);
      const T y2 = lerp_ref(e00, q[9], rc), y3 = lerp_ref(g0, g2, rc);
      const T r0 = lerp_ref(a00, e00, rr), r1 = lerp_ref(f0, g0, rr);
      const T r2 = lerp_ref(x2, y2, rr), r3 = lerp_ref(x3, y3, rr);
      const bool on[7] = {vf1,           vc1 && vf0,           vc1 && vf1,           odd_ok && vf0,
                          odd_ok && vf1, odd_ok && vc1 && vf0, odd_ok && vc1 && vf1};
      const T cv[7] = {on[0] ? a01 - f0 : (T)0,  on[1] ? q[0] - x2 : (T)0, on[2] ? q[1] - x3 : (T)0,
                       on[3] ? q[5] - r0 : (T)0, on[4] ? q[6] - r1 : (T)0, on[5] ? q[7] - r2 : (T)0,
                       on[6] ? q[8] - r3 : (T)0};
      const uint64_t c0 = (uint64_t)jc * A.dJ, c1 = (uint64_t)(mc + jc) * A.dJ, k1 = (uint64_t)mf;
      const uint64_t ub[7] = {ob_e + c0 + k1, ob_e + c1, ob_e + c1 + k1, ob_o + c0,
                              ob_o + c0 + k1, ob_o + c1, ob_o + c1 + k1};
      emit_uniform<T, OUT, 7>(A, cv, ub, (uint32_t)jf, on);
      a00 = q[2];
      a01 = q[3];
      a02 = q[4];
      f0 = f2;
      e00 = q[9];
      e02 = q[10];
      g0 = g2;
    }
  }
}

template <typename T, int OUT, int NV>
__device__ __forceinline__ void emit_values(const FusedArgs<T> &A, const T (&cv)[NV],
                                            const size_t (&lin)[NV], const bool (&on)[NV]) {
  if (OUT == OUT_T) {
#pragma unroll
    for (int k = 0; k < NV; k++)
      if (on[k]) A.coef[lin[k]] = cv[k];
  } else {
    emit_quantized<T, NV>(A, cv, lin, on);
  }
}

// The last cell column jf = mf - 1 (even f only: 3 of the 7 outputs): one thread per (R, jc).
template <typename T, int OUT>
__global__ void __launch_bounds__(256)
k_level_emit_lastcol(FusedArgs<T> A) {
  if (OUT == OUT_Q && A.qp) {
    A.quantizer = A.qp[A.level];
    A.volume = A.qp[A.nlev + A.level];
  }
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2;
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int total = mr * mc;
  // (uniform trip count: emit_quantized votes across the wave)
  for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {
    const int e = base + (int)threadIdx.x;
    const bool live = e < total;
    const int R = live ? e / mc : 0, jc = live ? e - (e / mc) * mc : 0;
    const int Pc0 = 2 * jc;
    const size_t of = (size_t)(nf - 1);  // padded position Pmax_f
    const T *pe0 = A.u + (size_t)min(2 * R, nr - 1) * A.uI + of;
    const T *po = A.u + (size_t)min(2 * R + 1, nr - 1) * A.uI + of;
    const T *pe2 = A.u + (size_t)min(2 * R + 2, nr - 1) * A.uI + of;
    const size_t r0 = (size_t)min(Pc0, nc - 1) * A.uJ, r1 = (size_t)min(Pc0 + 1, nc - 1) * A.uJ,
                 r2 = (size_t)min(Pc0 + 2, nc - 1) * A.uJ;
    const T a00 = pe0[r0], b0 = pe0[r1], c0v = pe0[r2];
    const T o00 = po[r0], o10 = po[r1];
    const T e00 = pe2[r0], e20 = pe2[r2];
    const T rc = Pc0 < nc ? A.ratio[1][Pc0] : (T)0;
    const T rr = 2 * R < nr ? A.ratio[0][2 * R] : (T)0;
    const bool vc1 = Pc0 + 1 <= Pmax_c && Pc0 + 1 != ghost_c;
    const bool odd_ok = 2 * R + 1 <= Pmax_r && 2 * R + 1 != ghost_r;
    const T x2 = lerp_ref(a00, c0v, rc), y2 = lerp_ref(e00, e20, rc);
    const T q0 = lerp_ref(a00, e00, rr), q2 = lerp_ref(x2, y2, rr);
    const bool on[3] = {live && vc1, live && odd_ok, live && odd_ok && vc1};
    const T cv[3] = {on[0] ? b0 - x2 : (T)0, on[1] ? o00 - q0 : (T)0, on[2] ? o10 - q2 : (T)0};
    const size_t ob_e = (size_t)R * A.dI, ob_o = (size_t)(mr + R) * A.dI;
    const size_t c0 = (size_t)jc * A.dJ, c1 = (size_t)(mc + jc) * A.dJ, k0 = (size_t)(mf - 1);
    const size_t lin[3] = {ob_e + c1 + k0, ob_o + c0 + k0, ob_o + c1 + k0};
    emit_values<T, OUT, 3>(A, cv, lin, on);
  }
}

} // namespace mgh
