// Decompression side, marching node restore (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused2.hpp"
#include "kernels_recompose.hpp"

namespace mgh {

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
  int zunused;
  slice_batch(A, &zunused);
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
  // Streaming stores for the reconstructed nodes: the output is not read again by this call, and
  // lines left dirty in the memory-side cache are written back at the expense of the quantized
  // integers still streaming in (node restore of all levels 395 -> 366 us at 512^3, one box,
  // alternating runs; nontemporal LOADS of the integers: slower, 1.07 -> 1.11 ms).
  auto store_pair = [](T *p, T a, T b) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    V2 v;
    v[0] = a;
    v[1] = b;
    __builtin_nontemporal_store(v, reinterpret_cast<V2 *>(p));
  };
  // one fine plane (real index rp) out: the four node values of the cell
  auto store_plane = [&](int rp, const T(&val)[4]) {
    T *pl = A.fine + (size_t)rp * A.fI;
    T *rowE = pl + outE, *rowO = pl + outO;
    if (vfo && (reinterpret_cast<uintptr_t>(rowE) & (2 * sizeof(T) - 1)) == 0) {
      store_pair(rowE, val[0], val[1]);
    } else {
      rowE[0] = val[0];
      if (vfo) rowE[1] = val[1];
    }
    if (vco) {
      if (vfo && (reinterpret_cast<uintptr_t>(rowO) & (2 * sizeof(T) - 1)) == 0) {
        store_pair(rowO, val[2], val[3]);
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
