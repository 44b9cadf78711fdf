// Decompression side, second generation of the load-vector pass (gfx950).
//
// k_level_loadvec_q (kernels_recompose.hpp) has the structure of the first-generation level kernel:
// one plane per step, two barriers per plane, every window load in its own predicated basic block.
// This one is the mirror of kernels_fused2.hpp:
//   * planes in (odd, even) PAIRS, two barriers per pair;
//   * the window of quantized coefficients is loaded UNCONDITIONALLY from clamped positions of the
//     reordered array, one pair ahead of its use (all loads of a pair in flight together), and
//     positions that carry no coefficient of this level -- outside the grid, ghost nodes, the
//     coarse corner of an even plane -- become 0 by a select on a lane-constant mask;
//   * sweep constants in LDS, r-sweep with tb(R+1) = td(R), r-chunks of a run-time length with
//     the last one taking what is left, tiles handed to the XCDs in contiguous ranges.
// Per value the operations and their order are those of k_level_loadvec_q (dequantize:
// LinearQuantization.hpp:246-264; sweeps: LPKFunctor.h:77-93) -- bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused2.hpp"
#include "kernels_recompose.hpp"

namespace mgh {

template <int TC, int TF, int RCH> struct Loadvec2Geom {
  static constexpr int WC = 2 * TC + 3, WF = 2 * TF + 3, HF = TF + 2, ROW = 2 * HF, PL = WC * ROW;
  static constexpr int TP = TF + 1;
  static constexpr int o_t1 = 2 * PL, o_wr = (o_t1 + 2 * WC * TP + 3) / 4 * 4, o_wf = o_wr + (RCH + 1) * 12,
                       o_wc = o_wf + 9 * TF, elems = o_wc + 9 * TC;
};

struct Loadvec2Grid {
  int gxm, ntile;   // tiles of TC x TF coarse nodes: gxm along f, ntile in all
  int rch, nchunk;  // r-chunks of rch <= RCH coarse planes, the last one takes what is left (<= rch + 1)
  int xcd_ranges;   // grid.x padded to a multiple of 8, tiles in contiguous ranges per XCD
};

template <typename T, typename QT, int TC, int TF, int RCH>
__global__ void __launch_bounds__(TC * TF)
k_level_loadvec2_q(RecomposeArgs<T> A, Loadvec2Grid G) {
  using GM = Loadvec2Geom<TC, TF, RCH>;
  constexpr int WC = GM::WC, WF = GM::WF, HF = GM::HF, ROW = GM::ROW, PL = GM::PL, TP = GM::TP;
  constexpr int NT = TC * TF;
  constexpr int BX = (WC * TF - 2 * NT);  // f-sweep items of the third round (per plane)
  static_assert(BX >= 0 && BX <= NT && BX % TF == 0, "f-sweep: two full rounds + one partial");
  __shared__ __attribute__((aligned(16))) T lds[GM::elems];
  T *const Cs0 = lds, *const Cs1 = lds + PL;                        // coefficient fields: odd / even plane
  T *const t1s0 = lds + GM::o_t1, *const t1s1 = t1s0 + WC * TP;      // f-swept rows of the pair
  T *const wrs = lds + GM::o_wr, *const wfs = lds + GM::o_wf, *const wcs = lds + GM::o_wc;
#define LI(lc, lf) ((lc) * ROW + ((lf) & 1) * HF + ((lf) >> 1))
  int b = blockIdx.x;
  if (G.xcd_ranges) {
    const int per = gridDim.x / 8;
    b = (b % 8) * per + b / 8;
    if (b >= G.ntile) return;
  }
  const int chunk = G.nchunk - 1 - (int)blockIdx.y;
  const int tid = threadIdx.x;
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int F0 = (b % G.gxm) * TF, C0 = (b / G.gxm) * TC, R0 = chunk * G.rch;
  const int rch = chunk == G.nchunk - 1 ? mr - R0 : G.rch;
  const int c_lo = 2 * C0 - 2, f_lo = 2 * F0 - 2, r_lo = 2 * R0 - 2;
  const int r_hi = min(2 * R0 + 2 * rch, 2 * mr);
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2, Pmax_f = 2 * mf - 2;
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int ghost_f = (nf % 2 == 0) ? nf - 1 : -7;

  for (int e = tid; e < rch * 9; e += NT) {
    const int R = R0 + e / 9, k = e % 9;
    wrs[(e / 9) * 12 + k] = R < mr ? A.mass[0][k * mr + R] : (T)0;
  }
  const int jf = tid % TF, jc = tid / TF;
  for (int e = tid; e < 9 * TF; e += NT) {
    const int k = e / TF, J = F0 + e % TF;
    wfs[e] = J < mf ? A.mass[2][k * mf + J] : (T)0;
  }
  for (int e = tid; e < 9 * TC; e += NT) {
    const int k = e / TC, J = C0 + e % TC;
    wcs[e] = J < mc ? A.mass[1][k * mc + J] : (T)0;
  }

  // ---- window elements of this thread: LDS index, offset inside an r-plane of the reordered
  // array (clamped: always a valid address), does the position carry a coefficient ------------
  constexpr int NL = (WC * WF + NT - 1) / NT;
  uint32_t lidx[NL], qoff[NL];
  bool okm[NL], evn[NL];
  auto reordered = [](int P, int m) { return (P & 1) ? m + (P - 1) / 2 : P / 2; };
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const int e = min(tid + k * NT, WC * WF - 1);
    const int lc = e / WF, lf = e - lc * WF;
    const int Pc = c_lo + lc, Pf = f_lo + lf;
    okm[k] = Pc >= 0 && Pc <= Pmax_c && Pc != ghost_c && Pf >= 0 && Pf <= Pmax_f && Pf != ghost_f;
    evn[k] = !(lc & 1) && !(lf & 1);
    int Pca = min(max(Pc, 0), Pmax_c), Pfa = min(max(Pf, 0), Pmax_f);
    if (Pca == ghost_c) Pca--;
    if (Pfa == ghost_f) Pfa--;
    lidx[k] = LI(lc, lf);
    qoff[k] = (uint32_t)reordered(Pca, mc) * (uint32_t)A.dJ + (uint32_t)reordered(Pfa, mf);
  }
  using QR = typename QReg<T, QT>::type;
  const QT *const src = qsrc<T>(A, QT()) + A.lin_base;
  auto plane_ok = [&](int p) { return p >= 0 && p <= Pmax_r && p != ghost_r; };
  auto plane_off = [&](int p) {
    int pa = min(max(p, 0), Pmax_r);
    if (pa == ghost_r) pa--;
    return (size_t)reordered(pa, mr) * A.dI;
  };
  // (an even plane holds no coefficient at its (even c, even f) positions -- a quarter of its
  // window: those lanes read the first element of the plane instead, one line that stays hot,
  // so that no memory traffic is spent on values the mask drops)
  auto fetch = [&](int p, QR(&reg)[NL]) {
    const QT *base = src + plane_off(p);
    const bool all = (p & 1) || A.allcoef;
#pragma unroll
    for (int k = 0; k < NL; k++) reg[k] = qload<T>(A, base + ((all || !evn[k]) ? qoff[k] : 0u));
  };
  // dequantized coefficient field of plane p into cs (0 where the position carries no
  // coefficient of this level: an even plane holds none at its (even c, even f) nodes)
  auto put = [&](int p, const QR(&reg)[NL], T *cs) {
    const bool pv = plane_ok(p);
    const bool all = (p & 1) || A.allcoef;
    const size_t pb = A.lin_base + plane_off(p);  // (index of a looked-up 16-bit symbol)
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const T v = qdecode(A, reg[k], pb + qoff[k]);
      cs[lidx[k]] = (pv && okm[k] && (all || !evn[k])) ? v : (T)0;
    }
  };

  auto f_sweep_row = [&](const T *cs, T *t1, int lc) {
    const T *row = cs + lc * ROW;
    const T a = row[jf], bq = row[HF + jf], c = row[jf + 1], d = row[HF + jf + 1], e = row[jf + 2];
    T wf[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wf[k] = wfs[k * TF + jf];
    const T tb = mass_tb(a, bq, c, wf);
    T tc = mass_tc(bq, c, d, wf);
    const T td = mass_td(c, d, e, wf);
    tc += tb * wf[7] + td * wf[8];
    t1[lc * TP + jf] = tc;
  };
  auto phase_b = [&](const T *cs, T *t1) {
    f_sweep_row(cs, t1, jc);
    f_sweep_row(cs, t1, jc + TC);
    if (BX > 0 && jc >= TC - BX / TF) f_sweep_row(cs, t1, jc + 2 * TC - (TC - BX / TF));
  };
  auto c_sweep = [&](const T *t1) {
    const T *col = t1 + (2 * jc) * TP + jf;
    const T a = col[0], bq = col[TP], c = col[2 * TP], d = col[3 * TP], e = col[4 * TP];
    T wc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wc[k] = wcs[k * TC + jc];
    const T tb = mass_tb(a, bq, c, wc);
    T tc = mass_tc(bq, c, d, wc);
    const T td = mass_td(c, d, e, wc);
    tc += tb * wc[7] + td * wc[8];
    return tc;
  };

  // ---- march: plane r_lo (even) on its own, then pairs (odd, even) --------------------------
  QR Po[NL], Pe[NL];
  fetch(r_lo, Pe);
  put(r_lo, Pe, Cs1);
  fetch(r_lo + 1, Po);
  fetch(r_lo + 2, Pe);
  __syncthreads();
  phase_b(Cs1, t1s1);
  __syncthreads();
  T e_prev = c_sweep(t1s1), o_prev = 0, td_prev = 0;
  const bool store_ok = C0 + jc < mc && F0 + jf < mf;
  const size_t load_off = (size_t)(C0 + jc) * mf + (F0 + jf);
  for (int p = r_lo + 1; p < r_hi; p += 2) {
    put(p, Po, Cs0);
    put(p + 1, Pe, Cs1);
    if (p + 2 < r_hi) {
      fetch(p + 2, Po);
      fetch(p + 3, Pe);
    }
    __syncthreads();
    phase_b(Cs0, t1s0);
    phase_b(Cs1, t1s1);
    __syncthreads();
    const T vo = c_sweep(t1s0);
    const T ve = c_sweep(t1s1);
    if (p + 1 == 2 * R0) {
      // first pair of the chunk: planes 2R0-2, 2R0-1, 2R0 give tb of coarse plane R0
      td_prev = e_prev * wrs[0] + vo * wrs[1] + ve * wrs[2];
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
        if (store_ok) A.load[(size_t)R * mc * mf + load_off] = tc;
      }
    }
    o_prev = vo;
    e_prev = ve;
  }
#undef LI
}

// ---------------------------------------------------------------------------------------------
// Node restore, marching: fine nodal array from the corrected coarse nodes and the quantized
// coefficients (GpkRev3D, GridProcessingKernel3D.hpp:1231-2352; dequantizer of
// LinearQuantization.hpp:246-264 fused in).
//
// k_level_restore2_q hands one wave a pair of fine rows of ONE plane: the four (node, f, c, fc)
// interpolants of a coarse plane are formed again for the odd plane below and the odd plane above
// it, and every wave pays the row set-up. Here a thread owns one CELL column -- coarse node
// (Jc, Jf) and its three odd neighbours -- and marches along r over a chunk of coarse planes,
// the mirror of the compression side's pair step (kernels_fused2.hpp): the interpolants G(R) of
// coarse plane R serve the even plane 2R directly and, r-lerped with G(R + 1), the odd plane
// 2R + 1; G(R + 1) is then carried on. 4 coarse loads, 12 lerps and 7 coefficient loads per cell
// and plane pair; no LDS, no barrier; a wave reads and writes whole 512-byte rows (64 cells of a
// row: 64 consecutive 8-byte coefficients of each parity class in, 64 consecutive node pairs
// out). Loads of the next pair are requested before the current one is finished.
// Padded coordinates as everywhere (even size: real last node at P = n, ghost at P = n - 1, which
// has no output). Per value the operations of k_level_restore_q in the same order.
//
// TODD (D = 4, an ODD slice of t): every node of the slice is a coefficient node:
// value = coefficient + lerp_t(X_a, X_b), X = the 3-D interpolant from the coarse slice below /
// above (CalcCoefficientsND.hpp:25-236).
// ---------------------------------------------------------------------------------------------
struct Restore3Grid {
  int gxm, ntile;   // cell tiles of TC x TF coarse nodes
  int rch, nchunk;  // coarse planes per workgroup; the last chunk takes what is left
};

template <typename T, typename QT, bool TODD, int TC, int TF>
__global__ void __launch_bounds__(TC * TF)
k_level_restore3_q(RecomposeArgs<T> A, Restore3Grid G) {
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int b = blockIdx.x;
  const int Jf = (b % G.gxm) * TF + (int)threadIdx.x % TF, Jc = (b / G.gxm) * TC + (int)threadIdx.x / TF;
  const int R0 = (int)blockIdx.y * G.rch;
  const int R1 = (int)blockIdx.y == G.nchunk - 1 ? mr : R0 + G.rch;
  if (Jc >= mc || Jf >= mf) return;
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2, Pmax_f = 2 * mf - 2;
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int ghost_f = (nf % 2 == 0) ? nf - 1 : -7;
  // the odd neighbours of the column exist?
  const bool vco = 2 * Jc + 1 <= Pmax_c && 2 * Jc + 1 != ghost_c;
  const bool vfo = 2 * Jf + 1 <= Pmax_f && 2 * Jf + 1 != ghost_f;
  const T rc = vco ? A.ratio[1][2 * Jc] : (T)0, rf = vfo ? A.ratio[2][2 * Jf] : (T)0;
  const int c1 = min(Jc + 1, mc - 1), f1 = min(Jf + 1, mf - 1);
  const uint32_t o00 = (uint32_t)(Jc * mf + Jf), o01 = (uint32_t)(Jc * mf + f1), o10 = (uint32_t)(c1 * mf + Jf),
                 o11 = (uint32_t)(c1 * mf + f1);
  const size_t mI = (size_t)mc * mf;
  // coefficient offsets inside an r-plane of the reordered array (clamped: always valid addresses)
  const uint32_t jo = (uint32_t)min(mc + Jc, nc - 1), ko = (uint32_t)min(mf + Jf, nf - 1);
  const uint32_t q_ee = (uint32_t)Jc * (uint32_t)A.dJ + (uint32_t)Jf, q_eo = (uint32_t)Jc * (uint32_t)A.dJ + ko,
                 q_oe = jo * (uint32_t)A.dJ + (uint32_t)Jf, q_oo = jo * (uint32_t)A.dJ + ko;
  // real positions of the cell's nodes in the output plane
  const int cE = min(2 * Jc, nc - 1), fE = min(2 * Jf, nf - 1);
  const uint32_t outE = (uint32_t)cE * (uint32_t)A.fJ + (uint32_t)fE;       // row E
  const uint32_t outO = (uint32_t)(2 * Jc + 1) * (uint32_t)A.fJ + (uint32_t)fE;  // row O (vco)
  const QT *const src = qsrc<T>(A, QT()) + A.lin_base;
  const ptrdiff_t to_b = TODD ? A.coarse_b - A.coarse : 0;
  const T rt = TODD ? A.ratio_t[A.tpos - 1] : (T)0;
  using QR = typename QReg<T, QT>::type;
  struct alignas(2 * sizeof(T)) Pair { T a, b; };

  // the four interpolants (node, f, c, fc) of coarse plane R at this column: f innermost, then c
  auto interp4 = [&](const T *cp, T(&Gv)[4]) {
    const T v00 = cp[o00], v01 = cp[o01], v10 = cp[o10], v11 = cp[o11];
    const T g0 = lerp_ref(v00, v01, rf), g1 = lerp_ref(v10, v11, rf);
    Gv[0] = v00;
    Gv[1] = g0;
    Gv[2] = lerp_ref(v00, v10, rc);
    Gv[3] = lerp_ref(g0, g1, rc);
  };
  struct Raw {  // what one coarse plane R needs from memory
    T v[TODD ? 8 : 4];   // coarse plane R + 1 (the planes of both neighbouring slices when TODD)
    QR qe[4], qo[4];     // coefficients of the even plane 2R and of the odd plane 2R + 1
  };
  auto plane_q = [&](int P) {  // reordered r index of padded plane P (clamped)
    int pa = min(P, Pmax_r);
    if (pa == ghost_r) pa--;
    return (size_t)((pa & 1) ? mr + (pa - 1) / 2 : pa / 2) * A.dI;
  };
  auto request = [&](int R, Raw &w) {
    const T *cp = A.coarse + (size_t)min(R + 1, mr - 1) * mI;
    w.v[0] = cp[o00];
    w.v[1] = cp[o01];
    w.v[2] = cp[o10];
    w.v[3] = cp[o11];
    if constexpr (TODD) {
      w.v[4] = cp[to_b + o00];
      w.v[5] = cp[to_b + o01];
      w.v[6] = cp[to_b + o10];
      w.v[7] = cp[to_b + o11];
    }
    const QT *qe = src + plane_q(2 * R), *qo = src + plane_q(2 * R + 1);
    w.qe[0] = qload<T>(A, qe + q_ee);
    w.qe[1] = qload<T>(A, qe + q_eo);
    w.qe[2] = qload<T>(A, qe + q_oe);
    w.qe[3] = qload<T>(A, qe + q_oo);
    w.qo[0] = qload<T>(A, qo + q_ee);
    w.qo[1] = qload<T>(A, qo + q_eo);
    w.qo[2] = qload<T>(A, qo + q_oe);
    w.qo[3] = qload<T>(A, qo + q_oo);
  };
  auto interp_from = [&](const T *v, T(&Gv)[4]) {
    const T g0 = lerp_ref(v[0], v[1], rf), g1 = lerp_ref(v[2], v[3], rf);
    Gv[0] = v[0];
    Gv[1] = g0;
    Gv[2] = lerp_ref(v[0], v[2], rc);
    Gv[3] = lerp_ref(g0, g1, rc);
  };
  // one fine plane (real index rp) out: the four node values of the cell
  auto store_plane = [&](int rp, const T(&val)[4]) {
    T *pl = A.fine + (size_t)rp * A.fI;
    T *rowE = pl + outE, *rowO = pl + outO;
    if (vfo && (reinterpret_cast<uintptr_t>(rowE) & (2 * sizeof(T) - 1)) == 0) {
      *reinterpret_cast<Pair *>(rowE) = Pair{val[0], val[1]};
    } else {
      rowE[0] = val[0];
      if (vfo) rowE[1] = val[1];
    }
    if (vco) {
      if (vfo && (reinterpret_cast<uintptr_t>(rowO) & (2 * sizeof(T) - 1)) == 0) {
        *reinterpret_cast<Pair *>(rowO) = Pair{val[2], val[3]};
      } else {
        rowO[0] = val[2];
        if (vfo) rowO[1] = val[3];
      }
    }
  };

  T Gp[4], Gpb[TODD ? 4 : 1];
  interp4(A.coarse + (size_t)R0 * mI, Gp);
  if constexpr (TODD) interp4(A.coarse + to_b + (size_t)R0 * mI, Gpb);
  Raw cur, nxt;
  request(R0, cur);
  for (int R = R0; R < R1; R++) {
    if (R + 1 < R1) request(R + 1, nxt);
    // ---- even plane P = 2R (real index min(2R, nr - 1)): the coarse node itself, three coefficients
    {
      const size_t pb = A.lin_base + plane_q(2 * R);
      T val[4];
      if constexpr (TODD) {
        val[0] = qdecode(A, cur.qe[0], pb + q_ee) + lerp_ref(Gp[0], Gpb[0], rt);
        val[1] = qdecode(A, cur.qe[1], pb + q_eo) + lerp_ref(Gp[1], Gpb[1], rt);
        val[2] = qdecode(A, cur.qe[2], pb + q_oe) + lerp_ref(Gp[2], Gpb[2], rt);
        val[3] = qdecode(A, cur.qe[3], pb + q_oo) + lerp_ref(Gp[3], Gpb[3], rt);
      } else {
        val[0] = Gp[0];
        val[1] = qdecode(A, cur.qe[1], pb + q_eo) + Gp[1];
        val[2] = qdecode(A, cur.qe[2], pb + q_oe) + Gp[2];
        val[3] = qdecode(A, cur.qe[3], pb + q_oo) + Gp[3];
      }
      store_plane(min(2 * R, nr - 1), val);
    }
    // ---- odd plane P = 2R + 1: r-lerp of the interpolants of the coarse planes R and R + 1
    T Gn[4], Gnb[TODD ? 4 : 1];
    interp_from(cur.v, Gn);
    if constexpr (TODD) interp_from(cur.v + 4, Gnb);
    const int P = 2 * R + 1;
    if (P <= Pmax_r && P != ghost_r) {
      const T rr = A.ratio[0][2 * R];
      const size_t pb = A.lin_base + plane_q(P);
      T val[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        T I = lerp_ref(Gp[k], Gn[k], rr);
        if constexpr (TODD) I = lerp_ref(I, lerp_ref(Gpb[k], Gnb[k], rr), rt);
        const uint32_t qo_off = k == 0 ? q_ee : (k == 1 ? q_eo : (k == 2 ? q_oe : q_oo));
        val[k] = qdecode(A, cur.qo[k], pb + qo_off) + I;
      }
      store_plane(P, val);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      Gp[k] = Gn[k];
      if constexpr (TODD) Gpb[k] = Gnb[k];
    }
    if (R + 1 < R1) cur = nxt;
  }
}

} // namespace mgh
