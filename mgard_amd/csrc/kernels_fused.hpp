// Fused level kernel for gfx950: one pass over the level-l nodal array produces
//   (a) the coarse nodal values (level l-1, before correction),
//   (b) the level-l multilevel coefficients, either as T in the reordered
//       layout or already quantized to int64 (+ outliers),
//   (c) the load vector  Lr(Lc(Lf(C)))  of the correction, i.e. the output of
//       the three fused mass-matrix/restriction sweeps,
// replacing CopyND + GpkReo3D + Lpk1/2/3Reo3D (+ the quantizer for this
// level's coefficients) of the reference
// (DataRefactoring.hpp:80-109, GridProcessingKernel3D.hpp:21-1179,
//  LinearProcessingKernel3D.hpp:27-1048, LinearQuantization.hpp:146-245).
//
// Work decomposition ("marching"): a 256-thread workgroup owns a tile of
// TC x TF coarse nodes in (c, f) and marches along r over RCH coarse planes.
// Everything is expressed in PADDED fine coordinates P in [0, 2m-2] per dim
// (m = coarse size): for an even-sized dim the real last node sits at P = n and
// P = n-1 is the ghost node whose coefficient is zero (Hierarchy.hpp:38-42,
// LinearProcessingKernel3D.hpp:52,177-203). For every fine plane the block
//   A. computes the coefficient field C on its (2TC+3) x (2TF+3) window from up
//      to three raw planes held in an LDS ring (interpolation order f, c, r),
//      writes the owned coefficients / coarse nodes to HBM,
//   B. applies the f-sweep  (window rows x TF),
//   C. applies the c-sweep  (TC x TF, one value per thread) and pushes it into
//      a 5-deep register window,
//   D. every second plane applies the r-sweep on the register window and
//      writes one plane of the load vector.
// Raw planes are read once per block: HBM read amplification is
// (2TC+3)(2TF+3)(2RCH+3) / (2TC 2TF 2RCH) (1.36 for 8 x 32 x 16), the halo
// re-reads mostly hit in L2. All arithmetic keeps the reference's operation
// order (no FMA contraction): results are bit-identical to kernels_v1.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_v1.hpp"

namespace mgh {

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

// PAIR = true runs the (odd, even) plane pair through each phase together: 3 barriers per pair
// instead of 5 and more independent work per phase, at the price of ~15 more VGPRs. It pays
// on the small levels, where a block's march is pure latency; on the big levels occupancy
// matters more (PAIR = false).
template <typename T, int OUT, int TC, int TF, int RCH, bool PAIR>
__global__ void __launch_bounds__(TC * TF)
k_level_fused(FusedArgs<T> A) {
  constexpr int WC = 2 * TC + 3;
  constexpr int WF = 2 * TF + 3;
  constexpr int HF = TF + 2;     // even-f slots of a window row (odd-f slots: TF + 1)
  constexpr int ROW = 2 * HF;    // LDS row: [0,HF) even f, [HF, HF+TF+1) odd f (stride-1 access)
  constexpr int NT = TC * TF;  // one owned cell / one c-sweep output per thread
  constexpr int NH = (TC + 2) * (TF + 2) - TC * TF;  // halo cells
  static_assert(NH <= NT, "halo cells are handled in one extra pass");
  __shared__ T raw[3][WC * ROW];
  __shared__ T Cs2[PAIR ? 2 : 1][WC * ROW];     // coefficient field (of the plane pair)
  __shared__ T t1s2[PAIR ? 2 : 1][WC][TF + 1];  // f-swept rows
  __shared__ T rfs[WF];
  __shared__ T rcs[WC];
  __shared__ T rrs[2 * RCH + 3];  // ratio_r[p - 1] of plane p = r_lo + index
  __shared__ T wrs[RCH][9];       // r-sweep constants of the chunk's coarse planes
#define LI(lc, lf) ((lc) * ROW + ((lf) & 1) * HF + ((lf) >> 1))

  if (OUT == OUT_Q && A.qp) {
    A.quantizer = A.qp[A.level];
    A.volume = A.qp[A.nlev + A.level];
  }
  const int tid = threadIdx.x;
  // r-chunks in reverse launch order: whatever ran before this kernel (the norm reduction,
  // the level above) leaves the END of the level's input in the memory-side cache
  const int F0 = blockIdx.x * TF, C0 = blockIdx.y * TC, R0 = (gridDim.z - 1 - blockIdx.z) * RCH;
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int c_lo = 2 * C0 - 2, f_lo = 2 * F0 - 2;
  const int r_lo = 2 * R0 - 2;
  const int r_hi = min(2 * R0 + 2 * RCH, 2 * mr);  // planes beyond 2mr-2 are empty anyway
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2, Pmax_f = 2 * mf - 2;
  // ghost (padded) positions of even-sized dims; -7 = none
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int ghost_f = (nf % 2 == 0) ? nf - 1 : -7;

  // interpolation ratios of the window (index = padded position of the left node)
  for (int e = tid; e < WF; e += NT) {
    const int P = f_lo + e;
    rfs[e] = (P >= 0 && P < nf) ? A.ratio[2][P] : (T)0;
  }
  for (int e = tid; e < WC; e += NT) {
    const int P = c_lo + e;
    rcs[e] = (P >= 0 && P < nc) ? A.ratio[1][P] : (T)0;
  }
  for (int e = tid; e < 2 * RCH + 3; e += NT) {
    const int P = r_lo + e - 1;  // left neighbour of plane r_lo + e
    rrs[e] = (P >= 0 && P < nr) ? A.ratio[0][P] : (T)0;
  }
  for (int e = tid; e < RCH * 9; e += NT) {
    const int R = R0 + e / 9, k = e % 9;
    wrs[e / 9][k] = R < mr ? A.mass[0][k * mr + R] : (T)0;
  }
  // per-thread sweep constants: f-sweep for jf = tid % TF, c-sweep for jc = tid / TF
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
  T win[5] = {0, 0, 0, 0, 0};  // c-swept planes p-4 .. p

  // ---- raw-plane streaming: per-thread window elements e = tid + k*256 ----------------
  constexpr int NL = (WC * WF + NT - 1) / NT;
  int lidx[NL];       // LDS index, or -1 past the window
  uint32_t goff[NL];  // min(Pc, nc-1) * uJ + min(Pf, nf-1)
  bool gval[NL];      // padded position inside the grid
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const int e = tid + k * NT;
    const int lc = e / WF, lf = e - lc * WF;
    const int Pc = c_lo + lc, Pf = f_lo + lf;
    lidx[k] = e < WC * WF ? LI(lc, lf) : -1;
    gval[k] = e < WC * WF && Pc >= 0 && Pc <= Pmax_c && Pf >= 0 && Pf <= Pmax_f;
    goff[k] = gval[k] ? (uint32_t)(min(Pc, nc - 1) * (int)A.uJ + min(Pf, nf - 1)) : 0u;
  }
  T amax = 0;  // OUT_NONE: abs-max over everything this thread reads (max is idempotent)
  auto fetch = [&](int p, T(&reg)[NL]) {
    const bool pv = p >= 0 && p <= Pmax_r;
    const T *base = A.u + (size_t)min(max(p, 0), nr - 1) * A.uI;
#pragma unroll
    for (int k = 0; k < NL; k++) reg[k] = (pv && gval[k]) ? base[goff[k]] : (T)0;
  };
  auto stash = [&](int p, const T(&reg)[NL]) {
    T *dst = raw[(p + 6) % 3];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      if (lidx[k] >= 0) dst[lidx[k]] = reg[k];
      if (OUT == OUT_NONE) {  // here, not in fetch(): the loads stay in flight until now
        const T a = abs_t(reg[k]);
        amax = a > amax ? a : amax;
      }
    }
  };

  // ---- cells: a cell is the 2x2 group of window nodes (lc0 + {0,1}, lf0 + {0,1}) with even
  // lc0, lf0. The thread's OWNED cell is (jc, jf); threads < NH also take one HALO cell. ------
  struct Cell {
    int lc0, lf0;      // window coordinates of the (even, even) node
    bool c1, f1;       // odd row / odd column of the cell lies inside the window
    bool vc0, vc1, vf0, vf1;  // node exists in the grid (and is not a ghost node)
    T rc, rf;          // interpolation ratios at the left (even) nodes
    int i00, i01, i02, i10, i11, i20, i22;  // LDS indices (clamped inside the window)
  };
  auto make_cell = [&](int cj, int fj) {
    Cell c;
    c.lc0 = 2 * cj + 2;
    c.lf0 = 2 * fj + 2;
    c.c1 = cj < TC;
    c.f1 = fj < TF;
    const int Pc0 = c_lo + c.lc0, Pf0 = f_lo + c.lf0;
    c.vc0 = Pc0 >= 0 && Pc0 <= Pmax_c;
    c.vf0 = Pf0 >= 0 && Pf0 <= Pmax_f;
    c.vc1 = c.c1 && Pc0 + 1 >= 0 && Pc0 + 1 <= Pmax_c && Pc0 + 1 != ghost_c;
    c.vf1 = c.f1 && Pf0 + 1 >= 0 && Pf0 + 1 <= Pmax_f && Pf0 + 1 != ghost_f;
    const int dc1 = c.c1 ? 1 : 0, dc2 = c.c1 ? 2 : 0, df1 = c.f1 ? 1 : 0, df2 = c.f1 ? 2 : 0;
    c.rc = rcs[c.lc0];
    c.rf = rfs[c.lf0];
    c.i00 = LI(c.lc0, c.lf0);
    c.i01 = LI(c.lc0, c.lf0 + df1);
    c.i02 = LI(c.lc0, c.lf0 + df2);
    c.i10 = LI(c.lc0 + dc1, c.lf0);
    c.i11 = LI(c.lc0 + dc1, c.lf0 + df1);
    c.i20 = LI(c.lc0 + dc2, c.lf0);
    c.i22 = LI(c.lc0 + dc2, c.lf0 + df2);
    return c;
  };
  __syncthreads();  // rfs / rcs visible
  const Cell own = make_cell(jc, jf);
  int hcj = 0, hfj = 0;
  if (tid < NH) {
    if (tid < 2 * (TF + 2)) {
      hcj = tid < TF + 2 ? -1 : TC;
      hfj = tid % (TF + 2) - 1;
    } else {
      const int h2 = tid - 2 * (TF + 2);
      hfj = h2 < TC ? -1 : TF;
      hcj = h2 % TC;
    }
  }
  const Cell halo = make_cell(hcj, hfj);
  // output offsets of the owned cell (reordered layout): c index C0+jc / mc+C0+jc, same in f
  const size_t ob_c0 = (size_t)(C0 + jc) * A.dJ, ob_c1 = (size_t)(mc + C0 + jc) * A.dJ;
  const int ok0 = F0 + jf, ok1 = mf + F0 + jf;
  const size_t coarse_off = (size_t)(C0 + jc) * mf + (F0 + jf);

  // coefficient field of one cell on plane p; returns the four values (ee, eo, oe, oo) and the
  // raw centre of the (even, even) node. Interpolation: f innermost, then c, then r
  // (GridProcessingKernel3D.hpp:614-617, 737-744, 854-871).
  auto cell_coeff = [&](const Cell &c, int p, bool pv, T rr, T *Cs, T(&cv)[4], T &centre) {
    const bool p_odd = p & 1;
    const T *cur = raw[(p + 6) % 3];
    T r[4];
    T v00, v01, v10, v11;
    if (!p_odd) {
      v00 = cur[c.i00];
      v01 = cur[c.i01];
      v10 = cur[c.i10];
      v11 = cur[c.i11];
      const T v02 = cur[c.i02], v20 = cur[c.i20], v22 = cur[c.i22];
      const T f0 = lerp_ref(v00, v02, c.rf), f2 = lerp_ref(v20, v22, c.rf);
      r[0] = v00;  // coarse node: no coefficient
      r[1] = f0;
      r[2] = lerp_ref(v00, v20, c.rc);
      r[3] = lerp_ref(f0, f2, c.rc);
    } else {
      const T *prv = raw[(p + 5) % 3];
      const T *nxt = raw[(p + 7) % 3];
      v00 = cur[c.i00];
      v01 = cur[c.i01];
      v10 = cur[c.i10];
      v11 = cur[c.i11];
      T g[2][4];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const T *pl = s2 ? nxt : prv;
        const T a00 = pl[c.i00], a02 = pl[c.i02], a20 = pl[c.i20], a22 = pl[c.i22];
        const T f0 = lerp_ref(a00, a02, c.rf), f2 = lerp_ref(a20, a22, c.rf);
        g[s2][0] = a00;
        g[s2][1] = f0;
        g[s2][2] = lerp_ref(a00, a20, c.rc);
        g[s2][3] = lerp_ref(f0, f2, c.rc);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) r[k] = lerp_ref(g[0][k], g[1][k], rr);
    }
    centre = v00;
    const bool m0 = pv && c.vc0 && c.vf0, m1 = pv && c.vc0 && c.vf1, m2 = pv && c.vc1 && c.vf0,
               m3 = pv && c.vc1 && c.vf1;
    cv[0] = (m0 && p_odd) ? v00 - r[0] : (T)0;
    cv[1] = m1 ? v01 - r[1] : (T)0;
    cv[2] = m2 ? v10 - r[2] : (T)0;
    cv[3] = m3 ? v11 - r[3] : (T)0;
    Cs[c.i00] = cv[0];
    if (c.f1) Cs[c.i01] = cv[1];
    if (c.c1) Cs[c.i10] = cv[2];
    if (c.c1 && c.f1) Cs[c.i11] = cv[3];
  };

  // Phase A for one fine plane whose raw neighbours are in the LDS ring: coefficient field of
  // the window into Cs, owned coefficients / coarse nodes to HBM
  auto phase_a = [&](int p, T *Cs) {
    const bool p_odd = p & 1;
    const bool pv = p >= 0 && p <= Pmax_r && p != ghost_r;
    const T rr = rrs[p - r_lo];
    T cv[4], centre;
    cell_coeff(own, p, pv, rr, Cs, cv, centre);
    const bool own_r = pv && p >= 2 * R0 && p < 2 * R0 + 2 * RCH;
    if (own_r) {
      const int oi = p_odd ? mr + (p - 1) / 2 : p / 2;
      const size_t ob = (size_t)oi * A.dI;
      const bool on[4] = {p_odd && own.vc0 && own.vf0, own.vc0 && own.vf1, own.vc1 && own.vf0,
                          own.vc1 && own.vf1};
      const size_t lin[4] = {ob + ob_c0 + ok0, ob + ob_c0 + ok1, ob + ob_c1 + ok0,
                             ob + ob_c1 + ok1};
      if (!p_odd && own.vc0 && own.vf0)
        A.coarse[(size_t)(p / 2) * mc * mf + coarse_off] = centre;
      if (OUT == OUT_T) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (on[k]) A.coef[lin[k]] = cv[k];
      } else if (OUT == OUT_Q) {
        emit_quantized<T, 4>(A, cv, lin, on);
      }
    }
    if (tid < NH) {
      T hv[4], hc;
      cell_coeff(halo, p, pv, rr, Cs, hv, hc);
    }
  };
  // Phase B: f-sweep of the window rows lc = jc, jc + TC, ... at coarse column jf
  auto phase_b = [&](const T *Cs, T(*t1s)[TF + 1]) {
    for (int lc = jc; lc < WC; lc += TC) {
      const T *row = Cs + lc * ROW;
      t1s[lc][jf] = mass_apply(row[jf], row[HF + jf], row[jf + 1], row[HF + jf + 1], row[jf + 2], wf);
    }
  };
  // Phase C: c-sweep, one value per thread, pushed into the register window
  auto phase_c = [&](const T(*t1s)[TF + 1]) {
    const int lc = 2 * jc + 2;
    const T v = mass_apply(t1s[lc - 2][jf], t1s[lc - 1][jf], t1s[lc][jf], t1s[lc + 1][jf],
                           t1s[lc + 2][jf], wc);
    win[0] = win[1];
    win[1] = win[2];
    win[2] = win[3];
    win[3] = win[4];
    win[4] = v;
  };
  // Phase D: r-sweep on the register window after even plane p = 2R + 2
  auto phase_d = [&](int p) {
    if (p >= 2 * R0 + 2) {
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
  };

  // ---- march: planes r_lo (even), then pairs (odd, even); the next pair's raw planes are
  // in flight (global -> registers) while the current pair is processed ----------------------
  T pre_o[NL], pre_e[NL];
  fetch(r_lo, pre_e);
  stash(r_lo, pre_e);
  fetch(r_lo + 1, pre_o);
  fetch(r_lo + 2, pre_e);
  __syncthreads();
  phase_a(r_lo, Cs2[0]);
  __syncthreads();
  phase_b(Cs2[0], t1s2[0]);
  __syncthreads();
  phase_c(t1s2[0]);
  for (int p = r_lo + 1; p < r_hi; p += 2) {
    // ring slots of planes p-3 and p-2 are free: their last readers (phase A of the previous
    // pair) are behind a barrier; phases B-D do not touch the ring
    stash(p, pre_o);
    stash(p + 1, pre_e);
    __syncthreads();
    if (p + 2 < r_hi) {
      fetch(p + 2, pre_o);
      fetch(p + 3, pre_e);
    }
    if (PAIR) {
      phase_a(p, Cs2[0]);
      phase_a(p + 1, Cs2[PAIR ? 1 : 0]);
      __syncthreads();
      phase_b(Cs2[0], t1s2[0]);
      phase_b(Cs2[PAIR ? 1 : 0], t1s2[PAIR ? 1 : 0]);
      __syncthreads();
      phase_c(t1s2[0]);
      phase_c(t1s2[PAIR ? 1 : 0]);
      phase_d(p + 1);
    } else {
      phase_a(p, Cs2[0]);
      __syncthreads();
      phase_b(Cs2[0], t1s2[0]);
      __syncthreads();
      phase_c(t1s2[0]);
      // (Cs2[0] was last read before the previous barrier; t1s2[0] is rewritten only after
      // the next one)
      phase_a(p + 1, Cs2[0]);
      __syncthreads();
      phase_b(Cs2[0], t1s2[0]);
      __syncthreads();
      phase_c(t1s2[0]);
      phase_d(p + 1);
    }
  }
  if (OUT == OUT_NONE && A.absmax_bits) {
    for (int off = 32; off > 0; off >>= 1) {
      const T o = __shfl_down(amax, off, 64);
      amax = o > amax ? o : amax;
    }
    if ((tid & 63) == 0) {
      // non-negative IEEE values order like their bit patterns; the plain load keeps the
      // tens of thousands of waves from queueing on one atomic once the maximum has settled
      unsigned long long bits;
      if (sizeof(T) == 4) bits = __float_as_uint((float)amax); else bits = __double_as_longlong((double)amax);
      if (bits > __atomic_load_n(A.absmax_bits, __ATOMIC_RELAXED)) atomicMax(A.absmax_bits, bits);
    }
  }
#undef LI
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
__device__ __forceinline__ void make_qparams_body(const QParamArgs<T> &P) {
  T norm;
  if (P.d_norm) {
    norm = *P.d_norm;
  } else if (P.s_is_inf) {
    const unsigned long long bits = *P.scalar;
    if (sizeof(T) == 4) norm = (T)__uint_as_float((unsigned)bits); else norm = (T)__longlong_as_double((long long)bits);
  } else {
    const double sum = __longlong_as_double((long long)*P.scalar);
    norm = (T)sum;
    if (sizeof(T) == 4) norm = P.normalize ? (T)sqrtf((float)(norm / (T)P.total)) : (T)sqrtf((float)norm);
    else norm = P.normalize ? (T)sqrt((double)(norm / (T)P.total)) : (T)sqrt((double)norm);
  }
  if (!P.d_norm && norm == 0) norm = sizeof(T) == 4 ? (T)1.1920928955078125e-7f : (T)2.220446049250313e-16;
  *P.norm_out = norm;
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
  for (int l = 0; l < P.nlev; l++) {
    T q = (T)(abs_tol / P.den[l]);
    q = 1.0f / q;
    P.qp[l] = q;
    P.qp[P.nlev + l] = P.vol[l];
  }
  if (P.reset_count) *P.reset_count = 0;
  if (P.zero_next) *P.zero_next = 0;
}

template <typename T> __global__ void k_make_qparams(QParamArgs<T> P) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  make_qparams_body<T>(P);
}

} // namespace mgh
