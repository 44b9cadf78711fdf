// Fused level kernel, second generation (gfx950): the same pass as kernels_fused.hpp --
// coarse nodes + level coefficients (quantized) + load vector Lr(Lc(Lf(C))) from ONE read of
// the level's nodal array -- rebuilt around its measured limit. rocprofv3 SQ counters on the
// first generation (512^3 f32): 263 VALU + 181 SALU + 33 LDS instructions per wave and fine
// plane, VALU busy 65-70 % of the kernel's time, HBM at 40 %: the kernel is bound by
// instruction issue, and only ~70 of those VALU instructions are arithmetic the algorithm asks
// for. What is different here:
//   * planes are processed in (odd, even) PAIRS and the even plane comes first: its four
//     interpolants (node, f, c, fc) are exactly the upper neighbours the odd plane's r-lerp
//     needs, and the lower neighbours are the ones kept in registers from the pair before --
//     8 instead of 20 lerps per cell and pair, 11 instead of 19 LDS reads, and the raw ring
//     shrinks from three planes to two;
//   * out-of-grid and ghost nodes are handled by DATA, not control flow: loads are
//     unconditional from clamped addresses (no valid coefficient ever reads a clamped node),
//     coefficient values of invalid nodes are zeroed by one select on a lane mask that is
//     constant for the thread, and only the stores of boundary tiles are predicated;
//   * every global access is wave-uniform base (SGPR pair) + 32-bit lane-constant offset;
//   * quantization in 32 bits: v_cvt_i32_f32 of copysign(0.5 + |t q|, t) saturates, so one
//     unsigned compare against the dictionary size finds the outliers (whose exact 64-bit
//     value is recomputed on the rare path), and the upper word of an in-dictionary symbol is 0;
//   * the r-sweep keeps tb(R+1) = td(R) (the same expression with the same operands:
//     LPKFunctor.h:77-93, dist[2R], dist[2R+1]) and a 3-value window instead of 5;
//   * 2 barriers per plane pair instead of 5: the next pair's raw planes are written into the
//     ring beside the f-sweep.
// Arithmetic per value is unchanged (same operations in the same order, no FMA contraction):
// results are bit-identical to kernels_fused.hpp / kernels_v1.hpp.
// Reference: DataRefactoring.hpp:80-109, GridProcessingKernel3D.hpp:21-1179,
// LinearProcessingKernel3D.hpp:27-1048, LinearQuantization.hpp:146-245.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace mgh {

// Developer build -DMGH_PHASE_TIMING: shader-clock time of the phases of a pair step, summed over
// the pairs of a few tiles (wave 0 carries the halo cells, wave 3 does not), read back with
// mgh_debug_phase_read (capi.hip).
#ifdef MGH_PHASE_TIMING
__device__ unsigned long long g_phase[2][8];
#define MGH_PT_DECL unsigned long long pt_t = clock64(); const bool pt_on = (blockIdx.x % 97 == 5) && (threadIdx.x == 0 || threadIdx.x == 192); const int pt_w = threadIdx.x == 0 ? 0 : 1;
#define MGH_PT(k) do { const unsigned long long pt_n = clock64(); if (pt_on) atomicAdd(&g_phase[pt_w][k], pt_n - pt_t); pt_t = pt_n; } while (0)
#else
#define MGH_PT_DECL
#define MGH_PT(k)
#endif

// (int) x with saturation -- exactly v_cvt_i32_f32 / v_cvt_i32_f64 (C's conversion is undefined
// out of range; the instruction is not)
__device__ __forceinline__ int32_t cvt_i32_sat(float x) {
  int32_t r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ int32_t cvt_i32_sat(double x) {
  int32_t r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// the three restricted mass rows of one coarse node (LPKFunctor.h:77-93)
template <typename T> __device__ __forceinline__ T mass_tb(T a, T b, T c, const T (&w)[9]) {
  return a * w[0] + b * w[1] + c * w[2];
}
template <typename T> __device__ __forceinline__ T mass_tc(T b, T c, T d, const T (&w)[9]) {
  return b * w[2] + c * w[3] + d * w[4];
}
template <typename T> __device__ __forceinline__ T mass_td(T c, T d, T e, const T (&w)[9]) {
  return c * w[4] + d * w[5] + e * w[6];
}

// LDS of one tile (elements of T): raw ring 2 planes, coefficient fields 2 planes, f-swept rows,
// ratios, r-sweep constants (RCH + 1 coarse planes: the last chunk of a level owns one more)
template <int TC, int TF, int RCH, bool TODD = false> struct Fused2Geom {
  static constexpr int WC = 2 * TC + 3, WF = 2 * TF + 3, HF = TF + 2, ROW = 2 * HF, PL = WC * ROW;
  static constexpr int TP = TF + 1;
  static constexpr int NRAW = TODD ? 4 : 2;  // odd t-slices (D = 4) also hold the even planes of
                                             // the two neighbouring slices
  static constexpr int o_cs = NRAW * PL, o_t1 = (NRAW + 2) * PL, o_rf = o_t1 + 2 * WC * TP, o_rc = o_rf + WF,
                       o_rr = o_rc + WC, o_wr = (o_rr + 2 * RCH + 5 + 3) / 4 * 4,
                       o_wf = o_wr + (RCH + 1) * 12, o_wc = o_wf + 9 * TF, elems = o_wc + 9 * TC;
};

// Outlier slots of a pair step, ONE request per workgroup (round 5). The slots of the single outlier
// list come from one address (~11 ns per atomic once they queue); with one request per wave and
// plane a field whose values leave the dictionary -- a tolerance below the data's noise -- turned
// the 0.9 ms step of 512^3 into 1.27 ms at 1.7 % outliers and 7.8 ms from 20 % on. Now a wave that
// finds up to kOutlierStashOf<T> outliers in a plane only COUNTS them while it stores the plane
// (cnt[wave * 2 + plane]) and leaves (value, output offset) pairs in LDS; behind the pair's second
// barrier wave 0 asks for the slots of the whole workgroup, publishes the base through LDS, and
// every wave writes its entries. More outliers than the stash holds in one wave and plane take
// the old road, one request for the wave. A pair step without outliers pays one LDS
// store per wave and plane and two 16-byte LDS loads.
// entries per wave and plane: floats all 256 values of a wave's plane (the per-wave request never
// runs), doubles 64 (the tile's LDS leaves room for no more at two workgroups per CU)
template <typename T> constexpr int kOutlierStashOf = sizeof(T) == 4 ? 256 : 64;
template <typename T> struct OutlierShared {
  unsigned *cnt;              // [8]: waves x planes of the pair
  unsigned long long *base;   // first slot of the workgroup's request
  unsigned *flag;             // pair sequence number `base` belongs to
  T *val;                     // [8][kOutlierStashOf<T>]: coefficient values
  uint32_t *off;              // [8][kOutlierStashOf<T>]: their offsets inside the output plane
};

// One tile: TC x TF coarse nodes at (C0, F0), marching over the coarse planes [R0, R0 + rch).
// c_end / f_end: coarse indices from which on the nodes belong to another tile of the launch
// (face tiles, below); lds: Fused2Geom<TC, TF, RCH, TODD>::elems elements, 16-byte aligned.
//
// TODD (D = 4, kernels below): the volume A.u is an ODD slice of the slowest dimension t. Its
// nodes are all coefficients, interpolated f, c, r and then t (CalcCoefficientsND.hpp:25-236:
// nested lerps, fastest dim innermost): value - lerp_t(X(t-1), X(t+1)) where X(s) are the
// interpolants of the even slice s completed through r -- the four cell interpolants of its even
// planes (`ua`, `ub`: the same planes of the neighbouring slices), r-lerped on odd planes. The odd
// planes of the neighbours are never read. `rt` = ratio_t at the left neighbour; out_base =
// element offset of the slice inside the output array (outlier indices are global).
template <typename T, int OUT, int TC, int TF, int RCH, bool TODD = false, bool AGG = false>
__device__ __forceinline__ void level_tile2(FusedArgs<T> &A, const int F0, const int C0,
                                            const int R0, const int rch, const int c_end,
                                            const int f_end, T *lds, const OutlierShared<T> &OS,
                                            const T *ua = nullptr, const T *ub = nullptr,
                                            const T rt = 0, const size_t out_base = 0) {
  using GM = Fused2Geom<TC, TF, RCH, TODD>;
  constexpr int WC = GM::WC;
  constexpr int WF = GM::WF;
  constexpr int HF = GM::HF;     // even-f slots of a window row (odd-f slots: TF + 1)
  constexpr int ROW = GM::ROW;   // LDS row: [0,HF) even f, [HF, HF+TF+1) odd f (stride-1 access)
  constexpr int PL = GM::PL;     // one window plane
  constexpr int NT = TC * TF;    // one owned cell / one c-sweep output per thread
  constexpr int NH = (TC + 2) * (TF + 2) - TC * TF;  // halo cells
  constexpr int TP = GM::TP;     // pitch of the f-swept rows
  constexpr int BX = (WC * TF - 2 * NT);  // f-sweep items of the third round (per plane)
  static_assert(NH <= NT, "halo cells are handled in one extra pass");
  static_assert(BX >= 0 && BX <= NT && BX % TF == 0, "f-sweep: two full rounds + one partial");
  T *const raw0 = lds, *const raw1 = lds + PL;             // odd / even plane of the pair
  T *const rawa = lds + 2 * PL, *const rawb = lds + 3 * PL;  // TODD: even plane of slices t-1, t+1
  T *const Cs0 = lds + GM::o_cs, *const Cs1 = Cs0 + PL;    // coefficient fields of the pair
  T *const t1s0 = lds + GM::o_t1, *const t1s1 = t1s0 + WC * TP;  // f-swept rows of the pair
  T *const rfs = lds + GM::o_rf;
  T *const rcs = lds + GM::o_rc;
  T *const rrs = lds + GM::o_rr;  // ratio_r[p - 1] of plane p = r_lo + index
  T *const wrs = lds + GM::o_wr;  // r-sweep constants of the chunk, [rch][12]
  T *const wfs = lds + GM::o_wf;  // f-sweep constants of the tile's coarse columns, [9][TF]
  T *const wcs = lds + GM::o_wc;  // c-sweep constants of the tile's coarse rows, [9][TC]
#define LI(lc, lf) ((lc) * ROW + ((lf) & 1) * HF + ((lf) >> 1))

  const int tid = threadIdx.x;
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int c_lo = 2 * C0 - 2, f_lo = 2 * F0 - 2;
  const int r_lo = 2 * R0 - 2;
  const int r_hi = min(2 * R0 + 2 * rch, 2 * mr);  // planes beyond 2mr-2 are empty anyway
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
  for (int e = tid; e < 2 * rch + 3; e += NT) {
    const int P = r_lo + e - 1;  // left neighbour of plane r_lo + e
    rrs[e] = (P >= 0 && P < nr) ? A.ratio[0][P] : (T)0;
  }
  for (int e = tid; e < rch * 9; e += NT) {
    const int R = R0 + e / 9, k = e % 9;
    wrs[(e / 9) * 12 + k] = R < mr ? A.mass[0][k * mr + R] : (T)0;
  }
  // sweep constants of the tile (f-sweep at jf = tid % TF, c-sweep at jc = tid / TF). They are
  // read from LDS where they are used: 125 instead of 152 VGPRs (f32), a fourth workgroup per CU.
  // (Measured neutral on the top level, which runs at the rate of a pure stream with its
  // read/write mix -- tools/micro/rw_mix.hip -- whatever the occupancy.)
  const int jf = tid % TF, jc = tid / TF;
  for (int e = tid; e < 9 * TF; e += NT) {
    const int k = e / TF, J = F0 + e % TF;
    wfs[e] = J < mf ? A.mass[2][k * mf + J] : (T)0;
  }
  for (int e = tid; e < 9 * TC; e += NT) {
    const int k = e / TC, J = C0 + e % TC;
    wcs[e] = J < mc ? A.mass[1][k * mc + J] : (T)0;
  }

  // ---- raw-plane streaming: window elements e = tid + k*NT, loaded from clamped positions
  // (a node outside the grid is never read by a valid coefficient) ---------------------------
  constexpr int NL = (WC * WF + NT - 1) / NT;
  uint32_t lidx[NL];  // LDS index
  uint32_t goff[NL];  // clamp(Pc) * uJ + clamp(Pf)
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const int e = min(tid + k * NT, WC * WF - 1);
    const int lc = e / WF, lf = e - lc * WF;
    const int Pc = min(max(c_lo + lc, 0), nc - 1), Pf = min(max(f_lo + lf, 0), nf - 1);
    lidx[k] = LI(lc, lf);
    goff[k] = (uint32_t)Pc * (uint32_t)A.uJ + (uint32_t)Pf;
  }
  auto fetch_from = [&](const T *vol, int p, T(&reg)[NL]) {
    const T *base = vol + (size_t)min(max(p, 0), nr - 1) * A.uI;
#pragma unroll
    for (int k = 0; k < NL; k++) reg[k] = base[goff[k]];
  };
  auto fetch = [&](int p, T(&reg)[NL]) { fetch_from(A.u, p, reg); };
  T amax = 0;  // OUT_NONE: abs-max over everything this thread reads (clamped loads repeat grid
               // values, and max is idempotent)
  auto stash = [&](T *dst, const T(&reg)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
      dst[lidx[k]] = reg[k];
      if (OUT == OUT_NONE) {
        const T a = abs_t(reg[k]);
        amax = a > amax ? a : amax;
      }
    }
  };

  // ---- cells: a cell is the 2x2 group of window nodes (lc0 + {0,1}, lf0 + {0,1}) with even
  // lc0, lf0. The thread's OWNED cell is (jc, jf); threads < NH also take one HALO cell. ------
  struct Cell {
    bool c1, f1;       // odd row / odd column of the cell lies inside the window
    bool m0, m1, m2, m3;  // node (ee, eo, oe, oo) exists in the grid and is not a ghost node
    bool s0, s1, s2, s3;  // ... and belongs to this tile (store predicate)
    T rc, rf;          // interpolation ratios at the left (even) nodes
    uint32_t i00, i01, i02, i10, i11, i20, i22;  // LDS indices (clamped inside the window)
  };
  auto make_cell = [&](int cj, int fj) {
    Cell c;
    const int lc0 = 2 * cj + 2, lf0 = 2 * fj + 2;
    c.c1 = cj < TC;
    c.f1 = fj < TF;
    const int Pc0 = c_lo + lc0, Pf0 = f_lo + lf0;
    const bool vc0 = Pc0 >= 0 && Pc0 <= Pmax_c;
    const bool vf0 = Pf0 >= 0 && Pf0 <= Pmax_f;
    const bool vc1 = c.c1 && Pc0 + 1 >= 0 && Pc0 + 1 <= Pmax_c && Pc0 + 1 != ghost_c;
    const bool vf1 = c.f1 && Pf0 + 1 >= 0 && Pf0 + 1 <= Pmax_f && Pf0 + 1 != ghost_f;
    c.m0 = vc0 && vf0;
    c.m1 = vc0 && vf1;
    c.m2 = vc1 && vf0;
    c.m3 = vc1 && vf1;
    const bool mine = C0 + cj < c_end && F0 + fj < f_end;
    c.s0 = c.m0 && mine;
    c.s1 = c.m1 && mine;
    c.s2 = c.m2 && mine;
    c.s3 = c.m3 && mine;
    const int dc1 = c.c1 ? 1 : 0, dc2 = c.c1 ? 2 : 0, df1 = c.f1 ? 1 : 0, df2 = c.f1 ? 2 : 0;
    c.rc = rcs[lc0];
    c.rf = rfs[lf0];
    c.i00 = LI(lc0, lf0);
    c.i01 = LI(lc0, lf0 + df1);
    c.i02 = LI(lc0, lf0 + df2);
    c.i10 = LI(lc0 + dc1, lf0);
    c.i11 = LI(lc0 + dc1, lf0 + df1);
    c.i20 = LI(lc0 + dc2, lf0);
    c.i22 = LI(lc0 + dc2, lf0 + df2);
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
  // boundary tiles predicate their stores; everywhere else every owned node exists
  // store predicates of the four outputs of a lane: (even c, even f) and (odd c, even f) of its
  // own cell, (even c, odd f) and (odd c, odd f) of the cell to its left
  bool st_on[4];
  {
    const Cell left = make_cell(jc, jf - 1);
    const bool mine = C0 + jc < c_end && F0 + jf < f_end;
    st_on[0] = own.s0;
    st_on[1] = left.m1 && mine;
    st_on[2] = own.s2;
    st_on[3] = left.m3 && mine;
  }
  const bool all_on = __syncthreads_and(st_on[0] && st_on[1] && st_on[2] && st_on[3]) != 0;

  // output offsets of the owned cell relative to the output plane (reordered layout:
  // c index C0+jc / mc+C0+jc, same in f); planes of < 2^29 elements (capi.hip: fused_ok)
  const uint32_t oc0 = (uint32_t)(C0 + jc) * (uint32_t)A.dJ, oc1 = (uint32_t)(mc + C0 + jc) * (uint32_t)A.dJ;
  // The odd-f coefficients a tile stores are those of the cells ONE TO THE LEFT of its own
  // (cell F0 + jf - 1; the leftmost lane's comes from the halo cell, whose coefficient field the
  // mass sweeps need anyway): in the reordered layout they start at column mf = 2^k + 1, and with
  // the own cell's the 512 (f32 / int64: 256 / 512; symbols: 128) bytes a wave writes per row
  // would start one element past a line boundary -- two partial lines per wave and row, each
  // written a second time by the neighbouring tile. Measured with the int64 output (one box,
  // alternating runs): top-level pass of 1024^3 3.55 -> 3.16 ms, step of 512^3 -22 us.
  const uint32_t of0 = (uint32_t)(F0 + jf), of1 = (uint32_t)(mf + F0 + jf - 1 + (F0 + jf == 0));
  const uint32_t off[4] = {oc0 + of0, oc0 + of1, oc1 + of0, oc1 + of1};
  const uint32_t coarse_off = (uint32_t)(C0 + jc) * (uint32_t)mf + (uint32_t)(F0 + jf);
  const T qz = A.quantizer, qv = A.volume;
  const uint32_t dict = (uint32_t)A.dict_size, half = (uint32_t)(A.dict_size / 2);

  // even plane of a cell: the four interpolants (node, f, c, fc) and its coefficients.
  // Interpolation: f innermost, then c (GridProcessingKernel3D.hpp:614-617, 737-744).
  auto cell_even = [&](const Cell &c, const T *rw, T *cs, bool pv, T(&E)[4], T(&cv)[4]) {
    const T v00 = rw[c.i00], v01 = rw[c.i01], v02 = rw[c.i02], v10 = rw[c.i10], v11 = rw[c.i11],
            v20 = rw[c.i20], v22 = rw[c.i22];
    const T f0 = lerp_ref(v00, v02, c.rf), f2 = lerp_ref(v20, v22, c.rf);
    E[0] = v00;
    E[1] = f0;
    E[2] = lerp_ref(v00, v20, c.rc);
    E[3] = lerp_ref(f0, f2, c.rc);
    cv[0] = (T)0;  // coarse node: no coefficient
    cv[1] = (pv && c.m1) ? v01 - E[1] : (T)0;
    cv[2] = (pv && c.m2) ? v10 - E[2] : (T)0;
    cv[3] = (pv && c.m3) ? v11 - E[3] : (T)0;
    cs[c.i00] = cv[0];
    if (c.f1) cs[c.i01] = cv[1];
    if (c.c1) cs[c.i10] = cv[2];
    if (c.c1 && c.f1) cs[c.i11] = cv[3];
  };
  // odd plane: r-lerp of the interpolants of the planes below (G) and above (E)
  // (GridProcessingKernel3D.hpp:854-871)
  auto cell_odd = [&](const Cell &c, const T *rw, T *cs, bool pv, T rr, const T(&G)[4],
                      const T(&E)[4], T(&cv)[4]) {
    const T v00 = rw[c.i00], v01 = rw[c.i01], v10 = rw[c.i10], v11 = rw[c.i11];
    cv[0] = (pv && c.m0) ? v00 - lerp_ref(G[0], E[0], rr) : (T)0;
    cv[1] = (pv && c.m1) ? v01 - lerp_ref(G[1], E[1], rr) : (T)0;
    cv[2] = (pv && c.m2) ? v10 - lerp_ref(G[2], E[2], rr) : (T)0;
    cv[3] = (pv && c.m3) ? v11 - lerp_ref(G[3], E[3], rr) : (T)0;
    cs[c.i00] = cv[0];
    if (c.f1) cs[c.i01] = cv[1];
    if (c.c1) cs[c.i10] = cv[2];
    if (c.c1 && c.f1) cs[c.i11] = cv[3];
  };

  // TODD: the four interpolants of an even plane of a neighbouring slice
  auto interp2 = [&](const Cell &c, const T *rw, T(&E)[4]) {
    const T v00 = rw[c.i00], v02 = rw[c.i02], v20 = rw[c.i20], v22 = rw[c.i22];
    const T f0 = lerp_ref(v00, v02, c.rf), f2 = lerp_ref(v20, v22, c.rf);
    E[0] = v00;
    E[1] = f0;
    E[2] = lerp_ref(v00, v20, c.rc);
    E[3] = lerp_ref(f0, f2, c.rc);
  };
  // TODD: coefficients of one plane of the odd slice from the interpolants of its neighbours
  auto cell_todd = [&](const Cell &c, const T *rw, T *cs, bool pv, const T(&Xa)[4],
                       const T(&Xb)[4], T(&cv)[4]) {
    const T v00 = rw[c.i00], v01 = rw[c.i01], v10 = rw[c.i10], v11 = rw[c.i11];
    cv[0] = (pv && c.m0) ? v00 - lerp_ref(Xa[0], Xb[0], rt) : (T)0;
    cv[1] = (pv && c.m1) ? v01 - lerp_ref(Xa[1], Xb[1], rt) : (T)0;
    cv[2] = (pv && c.m2) ? v10 - lerp_ref(Xa[2], Xb[2], rt) : (T)0;
    cv[3] = (pv && c.m3) ? v11 - lerp_ref(Xa[3], Xb[3], rt) : (T)0;
    cs[c.i00] = cv[0];
    if (c.f1) cs[c.i01] = cv[1];
    if (c.c1) cs[c.i10] = cv[2];
    if (c.c1 && c.f1) cs[c.i11] = cv[3];
  };
  // TODD, one cell and plane pair: Ga / Gb = interpolants of the neighbours' previous even plane
  auto pair_todd = [&](const Cell &c, bool pv_o, bool pv_e, T rr, T(&Ga)[4], T(&Gb)[4],
                       T(&cvo)[4], T(&cve)[4]) {
    T Ea[4], Eb[4], Xa[4], Xb[4];
    interp2(c, rawa, Ea);
    interp2(c, rawb, Eb);
    cell_todd(c, raw1, Cs1, pv_e, Ea, Eb, cve);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      Xa[k] = lerp_ref(Ga[k], Ea[k], rr);
      Xb[k] = lerp_ref(Gb[k], Eb[k], rr);
      Ga[k] = Ea[k];
      Gb[k] = Eb[k];
    }
    cell_todd(c, raw0, Cs0, pv_o, Xa, Xb, cvo);
  };

  // coefficients of one plane to HBM: (even c, even f) = c0 and (odd c, even f) = c2 of the own
  // cell from registers, the odd-f ones of the cell to the left from the coefficient field `cs`
  // (call it behind the barrier that follows phase A). K0 = 1: even plane (slot 0 is the coarse
  // node, stored by the caller); oi = index of the output plane in the reordered layout.
  constexpr bool kLists = AGG && (OUT == OUT_Q || OUT == OUT_QH);  // workgroup-wide slot requests
  constexpr int kOutlierStash = kOutlierStashOf<T>;
  auto emit = [&](const T *cs, const T c0, const T c2, int oi, int K0, int which) {
    const size_t ob = out_base + (size_t)oi * A.dI;
    const T cv[4] = {c0, cs[own.i01 - 1], c2, cs[own.i11 - 1]};
    const bool(&on)[4] = st_on;
    if (OUT == OUT_T) {
      T *o = A.coef + ob;
      if (all_on) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k >= K0) o[off[k]] = cv[k];
      } else {
        if (K0 == 0 && on[0]) o[off[0]] = cv[0];
        if (on[1]) o[off[1]] = cv[1];
        if (on[2]) o[off[2]] = cv[2];
        if (on[3]) o[off[3]] = cv[3];
      }
      return;
    }
    if (OUT != OUT_Q && OUT != OUT_QH) return;
    constexpr bool kFixed = OUT == OUT_QH;
    int32_t qs[4];
    bool slow = false;
    if (kFixed || A.prep_huffman) {
      bool ol[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k < K0) continue;
        const T a = (T)0.5 + abs_t(cv[k] * qz * qv);
        const uint32_t s = (uint32_t)cvt_i32_sat(copysign_t(a, cv[k])) + half;
        ol[k] = on[k] && s >= dict;
        qs[k] = ol[k] ? 0 : (int32_t)s;
        slow |= ol[k];
      }
      // out-of-dictionary values: counted and stashed here, written behind the pair's second
      // barrier with slots the whole workgroup asks for at once (flush_outliers;
      // LinearQuantization.hpp:208-241)
      unsigned deferred = 0;
      if (__any(slow)) {
        unsigned long long masks[4];
        unsigned total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          masks[k] = k >= K0 ? __ballot(ol[k]) : 0ull;
          total += __popcll(masks[k]);
        }
        const int lane = tid & 63;
        const bool stash = kLists && total <= (unsigned)kOutlierStash;
        unsigned long long base = 0;
        if (!stash) {  // a wave full of outliers: its own request, as before
          if (lane == 0) base = atomicAdd(A.outlier_count, (unsigned long long)total);
          base = __shfl(base, 0, 64);
        }
        const int slot0 = ((tid >> 6) * 2 + which) * kOutlierStash;
        unsigned before = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k >= K0 && ol[k]) {
            const unsigned rank = before + __popcll(masks[k] & ((1ULL << lane) - 1ULL));
            if (kLists && stash) {
              OS.val[slot0 + rank] = cv[k];
              OS.off[slot0 + rank] = off[k];
            } else {
              const unsigned long long o = base + rank;
              if (o < A.outlier_cap) {
                A.outlier_idx[o] = ob + off[k];
                A.outlier_val[o] = quantize_fast(cv[k], qz, qv) + A.dict_size / 2;
              }
            }
          }
          before += __popcll(masks[k]);
        }
        deferred = stash ? total : 0;
      }
      if (kLists && (tid & 63) == 0) OS.cnt[(tid >> 6) * 2 + which] = deferred;
      if (!kFixed && A.q16) {
        uint16_t *o = A.q16 + ob;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k >= K0 && (all_on || on[k])) o[off[k]] = (uint16_t)qs[k];
      } else {
        int64_t *o = A.q + ob;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k >= K0 && (all_on || on[k])) {
            // streaming store: the gigabyte of integers is not read again by this pass, and lines
            // left dirty in the memory-side cache are written back at the expense of whatever
            // streams next (measured: the norm pass of the following step 162 -> 110 us, this
            // kernel 412 -> 449 us, the step 1.014 -> 0.994 ms)
            __builtin_nontemporal_store((int64_t)(uint32_t)qs[k], &o[off[k]]);
          }
      }
    } else {
      // no dictionary: plain integers; |value| >= 2^31 (the conversion saturated) goes through
      // the 64-bit conversion
      if (kLists && (tid & 63) == 0) OS.cnt[(tid >> 6) * 2 + which] = 0;
      int64_t *o = A.q + ob;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k < K0) continue;
        const T a = (T)0.5 + abs_t(cv[k] * qz * qv);
        qs[k] = cvt_i32_sat(copysign_t(a, cv[k]));
        slow |= qs[k] == INT32_MAX || qs[k] == INT32_MIN;
      }
      if (__any(slow)) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k >= K0 && (all_on || on[k])) o[off[k]] = quantize_fast(cv[k], qz, qv);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k >= K0 && (all_on || on[k])) o[off[k]] = (int64_t)qs[k];
      }
    }
  };

  // The pair's out-of-dictionary values into the list (behind the pair's second barrier): oi_o / oi_e
  // = output planes of the pair's odd / even plane, K0e = first value of the even plane that is a
  // coefficient; seq = the pair's sequence number.
  auto flush_outliers = [&](int oi_o, int oi_e, unsigned seq) {
    const uint4 ca = *reinterpret_cast<const uint4 *>(OS.cnt), cb = *reinterpret_cast<const uint4 *>(OS.cnt + 4);
    const unsigned c[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
    unsigned tot = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) tot += c[k];
    if (tot == 0) return;  // (the same decision in every wave of the workgroup)
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
      *OS.base = atomicAdd(A.outlier_count, (unsigned long long)tot);
      __hip_atomic_store(OS.flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (lane 0 reads the base behind ITS acquire and hands it to the other lanes: a plain load of
    // theirs would have no happens-before edge to the leader's store)
    unsigned long long o = 0;
    if (lane == 0) {
      while (__hip_atomic_load(OS.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
      o = *OS.base;
    }
    o = __shfl(o, 0, 64);
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (k < 2 * wave) o += c[k];
#pragma unroll
    for (int which = 0; which < 2; which++) {
      const unsigned n = c[2 * wave + which];
      for (unsigned r = lane; r < n; r += 64) {
        const int e = (2 * wave + which) * kOutlierStash + (int)r;
        const unsigned long long slot = o + r;
        if (slot < A.outlier_cap) {
          A.outlier_idx[slot] = out_base + (size_t)(which ? oi_e : oi_o) * A.dI + OS.off[e];
          A.outlier_val[slot] = quantize_fast(OS.val[e], qz, qv) + A.dict_size / 2;
        }
      }
      o += n;
    }
  };

  // Phase B: f-sweep of the window rows of one plane at coarse column jf: rows jc, jc + TC,
  // and (threads of the last BX / TF values of jc) the remaining BX / TF rows
  auto f_sweep_row = [&](const T *cs, T *t1, int lc) {
    const T *row = cs + lc * ROW;
    const T a = row[jf], b = row[HF + jf], c = row[jf + 1], d = row[HF + jf + 1], e = row[jf + 2];
    T wf[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wf[k] = wfs[k * TF + jf];
    const T tb = mass_tb(a, b, c, wf);
    T tc = mass_tc(b, c, d, wf);
    const T td = mass_td(c, d, e, wf);
    tc += tb * wf[7] + td * wf[8];
    t1[lc * TP + jf] = tc;
  };
  auto phase_b = [&](const T *cs, T *t1) {
    f_sweep_row(cs, t1, jc);
    f_sweep_row(cs, t1, jc + TC);
    if (BX > 0 && jc >= TC - BX / TF) f_sweep_row(cs, t1, jc + 2 * TC - (TC - BX / TF));
  };
  // Phase C: c-sweep, one value per thread and plane
  auto c_sweep = [&](const T *t1) {
    const T *col = t1 + (2 * jc) * TP + jf;
    const T a = col[0], b = col[TP], c = col[2 * TP], d = col[3 * TP], e = col[4 * TP];
    T wc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wc[k] = wcs[k * TC + jc];
    const T tb = mass_tb(a, b, c, wc);
    T tc = mass_tc(b, c, d, wc);
    const T td = mass_td(c, d, e, wc);
    tc += tb * wc[7] + td * wc[8];
    return tc;
  };

  // ---- march ------------------------------------------------------------------------------
  T Go[4], Gh[4];  // interpolants of the previous even plane: owned cell, halo cell
  T e_prev;        // c-swept value of the previous even plane
  T td_prev = 0;   // r-sweep: td of the previous coarse plane = tb of the next one
  T Ga2[TODD ? 4 : 1], Gb2[TODD ? 4 : 1], Gha[TODD ? 4 : 1], Ghb[TODD ? 4 : 1];  // TODD
  // Raw planes travel global -> registers -> LDS ring, requested one pair ahead. (Two pairs
  // ahead -- two register sets, loop unrolled by two -- measured the same: 454 vs 459 us at
  // 172 VGPRs / 2 waves per SIMD; the pair step is not waiting for these loads.)
  struct Pre {
    T o[NL], e[NL], a[TODD ? NL : 1], b[TODD ? NL : 1];
  };
  Pre P0;
  auto fetch_pair = [&](int p, Pre &Q) {  // planes (p, p + 1) of the pair starting at odd p
    fetch(p, Q.o);
    fetch(p + 1, Q.e);
    if constexpr (TODD) {
      fetch_from(ua, p + 1, Q.a);
      fetch_from(ub, p + 1, Q.b);
    }
  };
  auto stash_pair = [&](const Pre &Q) {
    stash(raw0, Q.o);
    stash(raw1, Q.e);
    if constexpr (TODD) {
      stash(rawa, Q.a);
      stash(rawb, Q.b);
    }
  };
  fetch(r_lo, P0.e);
  stash(raw1, P0.e);
  if constexpr (TODD) {
    fetch_from(ua, r_lo, P0.a);
    fetch_from(ub, r_lo, P0.b);
    stash(rawa, P0.a);
    stash(rawb, P0.b);
  }
  fetch_pair(r_lo + 1, P0);
  __syncthreads();
  {
    const bool pv = r_lo >= 0 && r_lo <= Pmax_r && r_lo != ghost_r;
    T cv[4];
    if constexpr (TODD) {
      interp2(own, rawa, Ga2);
      interp2(own, rawb, Gb2);
      cell_todd(own, raw1, Cs1, pv, Ga2, Gb2, cv);
      if (tid < NH) {
        interp2(halo, rawa, Gha);
        interp2(halo, rawb, Ghb);
        cell_todd(halo, raw1, Cs1, pv, Gha, Ghb, cv);
      }
    } else {
      cell_even(own, raw1, Cs1, pv, Go, cv);
      if (tid < NH) cell_even(halo, raw1, Cs1, pv, Gh, cv);
    }
  }
  __syncthreads();
  // (the first pair's raw planes go into the ring next to the f-sweep of the first plane,
  // exactly like every later pair's)
  stash_pair(P0);
  if (r_lo + 3 < r_hi) fetch_pair(r_lo + 3, P0);
  phase_b(Cs1, t1s1);
  __syncthreads();
  e_prev = c_sweep(t1s1);
  T o_prev = 0;  // c-swept value of the previous odd plane
  // Two barriers per pair. Per pair and thread: A (reads the raw ring, writes Cs) | barrier |
  // stash of the NEXT pair's raw planes + f-sweep (reads Cs, writes t1s) + the pair's coefficients
  // to HBM (read Cs) | barrier | c- and r-sweep (read t1s) -- and straight on into A of the next pair: the ring was refilled before
  // the last barrier, Cs was last read before it, and t1s is rewritten only behind the next one.
  // Q holds the planes of the pair at p + 2 and is refilled with those of the pair at p + 4.
  auto pair_step = [&](const int p, Pre &Q) {
    MGH_PT_DECL
    // ---- phase A: coefficient fields of both planes ----
    const bool pv_o = p >= 0 && p <= Pmax_r && p != ghost_r;
    const bool pv_e = p + 1 >= 0 && p + 1 <= Pmax_r && p + 1 != ghost_r;
    T keep_o0, keep_o2, keep_e0, keep_e2;  // even-f coefficients of the own cell, stored behind the barrier
    {
      const T rr = rrs[p - r_lo];
      T E[4], cve[4], cvo[4];
      if constexpr (TODD) {
        pair_todd(own, pv_o, pv_e, rr, Ga2, Gb2, cvo, cve);
        if (tid < NH) {
          T ho[4], he[4];
          pair_todd(halo, pv_o, pv_e, rr, Gha, Ghb, ho, he);
        }
      } else {
        cell_even(own, raw1, Cs1, pv_e, E, cve);
        cell_odd(own, raw0, Cs0, pv_o, rr, Go, E, cvo);
#pragma unroll
        for (int k = 0; k < 4; k++) Go[k] = E[k];
        MGH_PT(6);
        if (pv_e && p + 1 < 2 * R0 + 2 * rch)
          if (all_on || own.s0) A.coarse[(size_t)((p + 1) / 2) * mc * mf + coarse_off] = E[0];
        MGH_PT(7);
        if (tid < NH) {
          T Eh[4], ch[4];
          cell_even(halo, raw1, Cs1, pv_e, Eh, ch);
          cell_odd(halo, raw0, Cs0, pv_o, rr, Gh, Eh, ch);
#pragma unroll
          for (int k = 0; k < 4; k++) Gh[k] = Eh[k];
        }
      }
      keep_o0 = cvo[0];
      keep_o2 = cvo[2];
      keep_e0 = cve[0];
      keep_e2 = cve[2];
    }
    MGH_PT(0);
    __syncthreads();
    MGH_PT(1);
    if (p + 2 < r_hi) {
      stash_pair(Q);
      if (p + 4 < r_hi) fetch_pair(p + 4, Q);
    }
    MGH_PT(2);
    phase_b(Cs0, t1s0);
    phase_b(Cs1, t1s1);
    // the planes' coefficients to HBM (odd-f ones from the field: see `of1`). Last in the phase:
    // in front of the ring refill or between refill and sweeps the top-level pass of 512^3 was
    // 10 us slower (one box, alternating runs)
    const bool em_o = pv_o && p >= 2 * R0, em_e = pv_e && p + 1 < 2 * R0 + 2 * rch;
    if (em_o) emit(Cs0, keep_o0, keep_o2, mr + (p - 1) / 2, 0, 0);
    else if (kLists && (tid & 63) == 0) OS.cnt[(tid >> 6) * 2] = 0;
    if (em_e) emit(Cs1, keep_e0, keep_e2, (p + 1) / 2, TODD ? 0 : 1, 1);
    else if (kLists && (tid & 63) == 0) OS.cnt[(tid >> 6) * 2 + 1] = 0;
    MGH_PT(3);
    __syncthreads();
    MGH_PT(4);
    if constexpr (kLists) flush_outliers(mr + (p - 1) / 2, (p + 1) / 2, (unsigned)(p + 64));
    // ---- phases C, D: c-sweep of both planes, r-sweep of coarse plane R = (p - 1) / 2 ----
    const T vo = c_sweep(t1s0);
    const T ve = c_sweep(t1s1);
    if (p + 1 == 2 * R0) {
      // first pair of the chunk: planes 2R0-2, 2R0-1, 2R0 give tb of coarse plane R0
      T wr[9];
#pragma unroll
      for (int k = 0; k < 3; k++) wr[k] = wrs[k];
      td_prev = e_prev * wr[0] + vo * wr[1] + ve * wr[2];
    } else {
      const int R = (p - 1) / 2;
      if (R < mr) {
        T wr[9];
#pragma unroll
        for (int k = 0; k < 9; k++) wr[k] = wrs[(R - R0) * 12 + k];
        T tc = mass_tc(o_prev, e_prev, vo, wr);
        const T td = mass_td(e_prev, vo, ve, wr);
        tc += td_prev * wr[7] + td * wr[8];
        td_prev = td;
        if (all_on || own.s0) A.load[(size_t)R * mc * mf + coarse_off] = tc;
      }
    }
    o_prev = vo;
    e_prev = ve;
    MGH_PT(5);
  };
  for (int p = r_lo + 1; p < r_hi; p += 2) pair_step(p, P0);
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

// Tiles of one launch. Sizes 2^k + 1 leave one coarse column / row / plane beyond the last full
// tile, and a whole tile for it costs almost as much as a full one (the instruction stream is the
// same): the remainder of f and c (up to 4 coarse nodes) goes to FACE tiles instead -- the same
// tile code instantiated 64 x 4 and 4 x 64 -- and the last r-chunk owns one plane more.
struct Fused2Grid {
  int gxm, n_main;    // main tiles TC x TF: gxm along f, n_main in all
  int ff_F0, n_ff;    // f-face: tiles of 64 x 4 at F0 = ff_F0, C0 = k * 64 (n_ff = 0: none)
  int cf_C0, n_cf;    // c-face: tiles of 4 x 64 at C0 = cf_C0, F0 = k * 64 (n_cf = 0: none)
  int rch, nchunk;    // r-chunks of rch <= RCH coarse planes; the last one takes what is left (<= rch + 1)
  int chunk_hi;       // this launch covers the chunks [chunk_hi - gridDim.y, chunk_hi) (a slab of the level)
  int xcd_ranges;     // tiles handed to the XCDs in contiguous ranges (grid.x padded to 8)
};

// D = 4: the level is processed slice by slice of the slowest dimension t with the 3-D tile code
// (the reference treats D > 3 "three dimensions at a time" as well: CalcCoefficientsND.hpp:25-236,
// CalcCorrectionND.hpp:131-155,199-213). blockIdx.z = slice index zi.
//   TMODE 1, even slices (padded position P = 2 zi, coarse index zi): exactly the 3-D pass of the
//     slice; coarse nodes -> slice zi of the 4-D coarse array, load vector -> slice P of the
//     per-slice load vectors (the t-sweep and the four solves follow in k_tsweep / ipk).
//   TMODE 2, odd slices (P = 2 zi + 1, coefficient index m_t + zi): TODD tiles.
template <typename T> struct Fused4 {
  const T *ratio_t;  // fine-level interpolation ratios of dim t
  size_t uT;         // element stride of t in the level's input
  size_t dT;         // element stride of t in the output array
  size_t cT;         // m_r * m_c * m_f
  int n_t, m_t;
};

template <typename T, int OUT, int TC, int TF, int RCH, bool FACES, int TMODE = 0, bool AGG = false>
__global__ void __launch_bounds__(TC * TF)
k_level_fused2(FusedArgs<T> A, Fused2Grid G, Fused4<T> Q) {
  static_assert(TC * TF == 256, "face tiles are 64 x 4 and 4 x 64");
  constexpr bool TODD = TMODE == 2;
  constexpr int e0 = Fused2Geom<TC, TF, RCH, TODD>::elems, e1 = Fused2Geom<64, 4, RCH, TODD>::elems,
                e2 = Fused2Geom<4, 64, RCH, TODD>::elems;
  constexpr int elems = FACES ? (e0 > e1 ? (e0 > e2 ? e0 : e2) : (e1 > e2 ? e1 : e2)) : e0;
  __shared__ __attribute__((aligned(16))) T lds[elems];
  constexpr bool kLists = AGG && (OUT == OUT_Q || OUT == OUT_QH);
  __shared__ __attribute__((aligned(16))) unsigned ol_cnt[8];
  __shared__ unsigned long long ol_base;
  __shared__ unsigned ol_flag;
  __shared__ T ol_val[kLists ? 8 * kOutlierStashOf<T> : 1];
  __shared__ uint32_t ol_off[kLists ? 8 * kOutlierStashOf<T> : 1];
  if (kLists && threadIdx.x == 0) ol_flag = 0;  // (no pair has sequence number 0; visible behind the tile's first barrier)
  const OutlierShared<T> OS{ol_cnt, &ol_base, &ol_flag, ol_val, ol_off};
  if ((OUT == OUT_Q || OUT == OUT_QH) && A.qinl) {
    inline_qparams<T>(A, threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0);
  } else if ((OUT == OUT_Q || OUT == OUT_QH) && A.qp) {
    A.quantizer = A.qp[A.level];
    A.volume = A.qp[A.nlev + A.level];
  }
  const T *ua = nullptr, *ub = nullptr;
  T rt = 0;
  size_t out_base = 0;
  if (TMODE != 0) {
    const int zi = blockIdx.z;
    const int P = 2 * zi + (TODD ? 1 : 0);            // padded position of the slice
    const T *vol = A.u;
    A.u = vol + (size_t)min(P, Q.n_t - 1) * Q.uT;     // (even n_t: the last node sits at P = n_t)
    A.load = A.load + (size_t)P * Q.cT;
    if (TODD) {
      ua = vol + (size_t)(P - 1) * Q.uT;
      ub = vol + (size_t)min(P + 1, Q.n_t - 1) * Q.uT;
      rt = Q.ratio_t[P - 1];
      out_base = (size_t)(Q.m_t + zi) * Q.dT;
    } else {
      A.coarse = A.coarse + (size_t)zi * Q.cT;
      out_base = (size_t)zi * Q.dT;
    }
  }
  // r-chunks in reverse launch order: whatever ran before this kernel (the norm reduction,
  // the level above) leaves the END of the level's input in the memory-side cache
  const int chunk = G.chunk_hi - 1 - (int)blockIdx.y;
  const int R0 = chunk * G.rch;
  const int rch = chunk == G.nchunk - 1 ? A.m[0] - R0 : G.rch;
  // workgroups go round-robin to the 8 XCDs (own L2 each): every XCD gets a contiguous range of
  // the launch's tiles, so that the partial cache lines neighbouring tiles write (the output rows
  // start at odd multiples of 8 bytes) meet in one L2. The grid is padded to a multiple of 8.
  // XCD k's range is [k n / 8, (k + 1) n / 8): the ranges differ by one tile at most (equal ranges of
  // ceil(n / 8) left the last XCD with what remained -- 33 tiles: 5, 5, 5, 5, 5, 5, 3, 0).
  int b = blockIdx.x;
  if (G.xcd_ranges) {
    const int n = G.n_main + G.n_ff + G.n_cf, k = b % 8, j = b / 8;
    if (G.xcd_ranges == 2) {  // (MGH_FUSED_XCD=2: the equal ranges of rounds 2-5, for A/B runs)
      b = k * (int)(gridDim.x / 8) + j;
      if (b >= n) return;
    } else {
      const int lo = k * n / 8, hi = (k + 1) * n / 8;
      if (j >= hi - lo) return;
      b = lo + j;
    }
  }
  const int f_main_end = G.n_ff ? G.ff_F0 : A.m[2], c_main_end = G.n_cf ? G.cf_C0 : A.m[1];
  if (!FACES || b < G.n_main) {
    level_tile2<T, OUT, TC, TF, RCH, TODD, AGG>(A, (b % G.gxm) * TF, (b / G.gxm) * TC, R0, rch,
                                           c_main_end, f_main_end, lds, OS, ua, ub, rt, out_base);
  } else if (b < G.n_main + G.n_ff) {
    level_tile2<T, OUT, 64, 4, RCH, TODD, AGG>(A, G.ff_F0, (b - G.n_main) * 64, R0, rch, A.m[1], A.m[2],
                                          lds, OS, ua, ub, rt, out_base);
  } else {
    level_tile2<T, OUT, 4, 64, RCH, TODD, AGG>(A, (b - G.n_main - G.n_ff) * 64, G.cf_C0, R0, rch, A.m[1],
                                          f_main_end, lds, OS, ua, ub, rt, out_base);
  }
}

// D = 4: quantize (or copy) the level-0 nodal values (compact (m0, m1, m2, m3)) into the head of
// the output; dT = element stride of t in the output array.
template <typename T, int OUT>
__global__ void __launch_bounds__(1024)
k_head_out4(int m0, int m1, int m2, int m3, const T *__restrict__ nodal, FusedArgs<T> A, size_t dT) {
  if ((OUT == OUT_Q || OUT == OUT_QH) && A.qp) {
    A.quantizer = A.qp[0];
    A.volume = A.qp[A.nlev];
  }
  __shared__ unsigned wcnt[16];
  __shared__ unsigned long long gbase;
  const int total = m0 * m1 * m2 * m3;
  // At level 0 every value is out of the dictionary, and the slots of the one outlier list come
  // from an atomicAdd on ONE address (~11 ns each once they queue): a whole 1024-thread workgroup
  // asks once per round -- 2 x 65^3 values of an 8 x 512^3 slab: 110 us with one atomic per value
  // or per wave, a few us like this.
  const int stride = gridDim.x * blockDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
  for (int e0 = blockIdx.x * blockDim.x; e0 < total; e0 += stride) {
    const int e = e0 + (int)threadIdx.x;
    const bool live = e < total;
    const int ee = live ? e : 0;
    const int k = ee % m3, j = (ee / m3) % m2, i = (ee / (m3 * m2)) % m1, t = ee / (m3 * m2 * m1);
    const size_t lin = (size_t)t * dT + (size_t)i * A.dI + (size_t)j * A.dJ + k;
    const T v = nodal[ee];
    if (OUT == OUT_T) {
      if (live) A.coef[lin] = v;
    } else {
      int64_t qd = quantize_one(v, A.quantizer, A.volume);
      bool ol = false;
      if (A.prep_huffman) {
        qd += A.dict_size / 2;
        ol = live && !(qd >= 0 && qd < A.dict_size);
      }
      const unsigned long long mask = __ballot(ol);
      if (lane == 0) wcnt[wave] = (unsigned)__popcll(mask);
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < nwave; w++) {
          const unsigned c = wcnt[w];
          wcnt[w] = tot;
          tot += c;
        }
        gbase = tot ? atomicAdd(A.outlier_count, (unsigned long long)tot) : 0ull;
      }
      __syncthreads();
      if (ol) {
        const unsigned long long o = gbase + wcnt[wave] + __popcll(mask & ((1ULL << lane) - 1ULL));
        if (o < A.outlier_cap) {
          A.outlier_idx[o] = lin;
          A.outlier_val[o] = qd;
        }
        qd = 0;
      }
      if (live) {
        if (A.q16) A.q16[lin] = (uint16_t)qd;
        else A.q[lin] = qd;
      }
      __syncthreads();  // (wcnt / gbase are rewritten by the next round)
    }
  }
}

// D = 4: the last mass/restriction sweep, along t, on the per-slice load vectors:
// L[P][j], P in [0, 2 m_t - 2] padded positions (a ghost slice is all zero), j < M = m_r m_c m_f
// -> out[T][j] (LPKFunctor.h:77-93; operands out of range are zero).
template <typename T>
__global__ void __launch_bounds__(256)
k_tsweep(const T *__restrict__ L, T *__restrict__ out, size_t M, int m_t, const T *__restrict__ mass) {
  const int Tt = blockIdx.y;
  T w[9];
#pragma unroll
  for (int k = 0; k < 9; k++) w[k] = mass[k * m_t + Tt];
  const int np = 2 * m_t - 1;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < M; j += (size_t)gridDim.x * 256) {
    T v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int P = 2 * Tt - 2 + k;
      v[k] = (P >= 0 && P < np) ? L[(size_t)P * M + j] : (T)0;
    }
    out[(size_t)Tt * M + j] = mass_apply(v[0], v[1], v[2], v[3], v[4], w);
  }
}

// The same with every input slice read ONCE: a thread owns one position j of the slices, takes
// its np = 2 m_t - 1 <= 2 MT - 1 values of L into registers and produces all m_t outputs (k_tsweep
// reads five input slices per output slice: 25 slice reads instead of 9 for m_t = 5).
template <typename T, int MT>
__global__ void __launch_bounds__(256)
k_tsweep_once(const T *__restrict__ L, T *__restrict__ out, size_t M, int m_t, const T *__restrict__ mass) {
  constexpr int NP = 2 * MT - 1;
  const int np = 2 * m_t - 1;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < M; j += (size_t)gridDim.x * 256) {
    T v[NP + 4];  // v[2 + P]: two zeros in front and behind (operands out of range are zero)
    v[0] = v[1] = (T)0;
#pragma unroll
    for (int P = 0; P < NP + 2; P++) v[2 + P] = (P < np) ? L[(size_t)min(P, np - 1) * M + j] : (T)0;
#pragma unroll
    for (int Tt = 0; Tt < MT; Tt++) {
      if (Tt < m_t) {
        T w[9];
#pragma unroll
        for (int k = 0; k < 9; k++) w[k] = mass[k * m_t + Tt];
        out[(size_t)Tt * M + j] = mass_apply(v[2 * Tt], v[2 * Tt + 1], v[2 * Tt + 2], v[2 * Tt + 3], v[2 * Tt + 4], w);
      }
    }
  }
}

// D = 4: the Thomas solve along t -- pencils of m_t <= MT values, one per thread, entirely in
// registers, every access coalesced across the threads -- with the correction applied to the
// coarse array on the way out (AddND / SubtractND): out[t][j] +/-= solve(corr[.][j])[t].
// tt = thomas table of dim t: [0,n) forward multiplier, [n,2n) am[i+1], [2n,3n) bm[i+1]
// (IPKFunctor.h:127,147; same expressions, same order as thomas_lds).
template <typename T, int MT>
__global__ void __launch_bounds__(256)
k_tsolve_apply(const T *__restrict__ corr, T *__restrict__ out, size_t M, int m_t,
               const T *__restrict__ tt, int sign) {
  T fw[MT], am[MT], bm[MT];
#pragma unroll
  for (int k = 0; k < MT; k++) {
    const int kk = k < m_t ? k : 0;
    fw[k] = tt[kk];
    am[k] = tt[m_t + kk];
    bm[k] = tt[2 * m_t + kk];
  }
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < M; j += (size_t)gridDim.x * 256) {
    T x[MT], o[MT];
#pragma unroll
    for (int k = 0; k < MT; k++)
      if (k < m_t) {
        x[k] = corr[(size_t)k * M + j];
        o[k] = out[(size_t)k * M + j];
      }
    T prev = 0;
#pragma unroll
    for (int k = 0; k < MT; k++)
      if (k < m_t) {
        x[k] = x[k] - prev * fw[k];
        prev = x[k];
      }
    prev = 0;
#pragma unroll
    for (int k = MT - 1; k >= 0; k--)
      if (k < m_t) {
        x[k] = (x[k] - am[k] * prev) / bm[k];
        prev = x[k];
      }
#pragma unroll
    for (int k = 0; k < MT; k++)
      if (k < m_t) out[(size_t)k * M + j] = sign > 0 ? o[k] + x[k] : o[k] - x[k];
  }
}

} // namespace mgh
