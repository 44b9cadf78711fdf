// Box level kernel (gfx950): the fused level pass for SMALL levels.
//
// kernels_fused2.hpp marches a tile along r: a chain of dependent phases per plane pair
// (~12 000 cycles per pair, profiles/r02*), which is the right shape where the march hides behind
// HBM bandwidth and the wrong one on the levels <= 129^3, whose time IS that chain: with one coarse
// plane per workgroup a level still costs prologue + two pair steps = ~18-24 us for a few hundred
// thousand nodes. Here nothing marches. A 256-thread workgroup owns a BOX of TR x TC x TF coarse
// nodes, loads the whole (2TR+3)(2TC+3)(2TF+3) window of fine nodes it needs in ONE round trip,
// and then every phase of the level runs once, over the whole box, separated by barriers:
//   window -> coefficient field -> f-sweep -> c-sweep -> r-sweep,
// five short phases instead of (2 RCH + 1) plane steps. Window and coefficient field are handled
// column-wise -- a thread owns one (c, f) position of the window and walks its 2TR+3 planes: no
// index division in any loop, the lerps of the even directions are computed and dropped by a
// select (no divergence), and the (f, c) interpolant of an even plane is computed once and serves
// the plane itself and the r-lerps of the two odd planes next to it, as in kernels_fused2.hpp.
// The owned coefficients are quantized and stored in output order beside the f-sweep.
// The halo is recomputed (2.2 x the owned nodes for a 4 x 4 x 8 box): all of it hits in L2, and
// these levels hold < 2 % of the data.
//
// Outliers: the small levels are where the out-of-dictionary values are (from 65^3 down nearly
// every coefficient), and slots of the one outlier list are handed out by an atomicAdd on ONE
// address, ~11 ns each once waves queue on it (measured: a level of 65^3 with one atomic per
// wave and 64 values = 4300 atomics = 46 us). Here a workgroup asks ONCE for the slots of all
// its 1024 owned values: wave counts meet in LDS at the barrier the f-sweep needs anyway, thread
// 0 adds, the waves write their entries behind the next barrier.
//
// Conventions are those of kernels_fused2.hpp: padded coordinates P in [0, 2m - 2] per dimension
// (even sizes: real last node at P = n, ghost at P = n - 1), out-of-grid and ghost nodes handled by
// DATA (clamped loads, coefficient forced to zero by a mask). Arithmetic per value is that of
// kernels_v1.hpp (gpk_reo_elem, lpk_elem): same operations in the same order, no FMA contraction
// -- bit-identical results.
// Reference: GridProcessingKernel3D.hpp:21-1179, LinearProcessingKernel3D.hpp:27-1048,
// LPKFunctor.h:77-93, LinearQuantization.hpp:146-245.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace mgh {

template <int TR, int TC, int TF> struct BoxGeom {
  static constexpr int WR = 2 * TR + 3, WC = 2 * TC + 3, WF = 2 * TF + 3;
  static constexpr int NW = WR * WC * WF;          // window nodes
  static constexpr int N1 = WR * WC * TF;          // f-swept values
  static constexpr int N2 = WR * TC * TF;          // c-swept values
  static constexpr int NO = 8 * TR * TC * TF;      // owned fine nodes
  static constexpr int o_cf = NW, o_t1 = 2 * NW, o_t2 = o_t1 + N1, o_rr = o_t2 + N2, o_rc = o_rr + WR,
                       o_rf = o_rc + WC, o_wr = o_rf + WF, o_wc = o_wr + 9 * TR, o_wf = o_wc + 9 * TC,
                       elems = o_wf + 9 * TF;
};

template <typename T, int OUT, int TR, int TC, int TF>
__global__ void __launch_bounds__(256)
k_level_box(FusedArgs<T> A, int gx, int gy) {
  using G = BoxGeom<TR, TC, TF>;
  constexpr int WR = G::WR, WC = G::WC, WF = G::WF, NT = 256, NCOL = WC * WF;
  constexpr bool kQuant = OUT == OUT_Q || OUT == OUT_QH;
  static_assert(NCOL <= NT, "one thread per (c, f) column of the window");
  static_assert(G::NO == 4 * NT && (TR * TC * TF) % 64 == 0, "four owned nodes per thread, whole waves per class");
  __shared__ __attribute__((aligned(16))) T lds[G::elems];
  __shared__ unsigned wcnt[4], wbase_lo[4];
  __shared__ unsigned long long gbase;
  T *const raw = lds, *const cf = lds + G::o_cf, *const t1 = lds + G::o_t1, *const t2 = lds + G::o_t2;
  T *const rrs = lds + G::o_rr, *const rcs = lds + G::o_rc, *const rfs = lds + G::o_rf;
  T *const wrs = lds + G::o_wr, *const wcs = lds + G::o_wc, *const wfs = lds + G::o_wf;
  if (kQuant && A.qp) {
    A.quantizer = A.qp[A.level];
    A.volume = A.qp[A.nlev + A.level];
  }
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int F0 = (b % gx) * TF, C0 = ((b / gx) % gy) * TC, R0 = (b / (gx * gy)) * TR;
  const int nr = A.n[0], nc = A.n[1], nf = A.n[2];
  const int mr = A.m[0], mc = A.m[1], mf = A.m[2];
  const int r_lo = 2 * R0 - 2, c_lo = 2 * C0 - 2, f_lo = 2 * F0 - 2;
  const int Pmax_r = 2 * mr - 2, Pmax_c = 2 * mc - 2, Pmax_f = 2 * mf - 2;
  const int ghost_r = (nr % 2 == 0) ? nr - 1 : -7;
  const int ghost_c = (nc % 2 == 0) ? nc - 1 : -7;
  const int ghost_f = (nf % 2 == 0) ? nf - 1 : -7;
  // the thread's column of the window
  const bool colthr = tid < NCOL;
  const int lc = colthr ? tid / WF : 0, lf = colthr ? tid % WF : 0;
  const int Pc = c_lo + lc, Pf = f_lo + lf;

  // ---- phase 0: window of fine nodes (clamped positions) and the tables of the box ----------
  {
    T reg[WR];
    if (colthr) {
      const T *col = A.u + (size_t)min(max(Pc, 0), nc - 1) * A.uJ + min(max(Pf, 0), nf - 1);
#pragma unroll
      for (int lr = 0; lr < WR; lr++) reg[lr] = col[(size_t)min(max(r_lo + lr, 0), nr - 1) * A.uI];
    }
    // interpolation ratios at the window positions (index = padded position of the LEFT node)
    for (int e = tid; e < WR; e += NT) {
      const int P = r_lo + e;
      rrs[e] = (P >= 0 && P < nr) ? A.ratio[0][P] : (T)0;
    }
    for (int e = tid; e < WC; e += NT) {
      const int P = c_lo + e;
      rcs[e] = (P >= 0 && P < nc) ? A.ratio[1][P] : (T)0;
    }
    for (int e = tid; e < WF; e += NT) {
      const int P = f_lo + e;
      rfs[e] = (P >= 0 && P < nf) ? A.ratio[2][P] : (T)0;
    }
    for (int e = tid; e < 9 * TR; e += NT) {
      const int k = e / TR, J = R0 + e % TR;
      wrs[e] = J < mr ? A.mass[0][k * mr + J] : (T)0;
    }
    for (int e = tid; e < 9 * TC; e += NT) {
      const int k = e / TC, J = C0 + e % TC;
      wcs[e] = J < mc ? A.mass[1][k * mc + J] : (T)0;
    }
    for (int e = tid; e < 9 * TF; e += NT) {
      const int k = e / TF, J = F0 + e % TF;
      wfs[e] = J < mf ? A.mass[2][k * mf + J] : (T)0;
    }
    if (colthr) {
#pragma unroll
      for (int lr = 0; lr < WR; lr++) raw[lr * NCOL + tid] = reg[lr];
    }
  }
  __syncthreads();

  // ---- phase 1: coefficient field of the thread's column. Node of parity (ro, co, fo): value
  // minus the nested interpolant of its even neighbours, f innermost, then c, then r
  // (GridProcessingKernel3D.hpp:614-617, 737-744, 854-871). G(plane) = the (f, c) interpolant of
  // an even plane at this column. All-even nodes, nodes outside the grid and ghost nodes carry 0.
  if (colthr) {
    const int co = lc & 1, fo = lf & 1;  // (the window starts at an even position)
    const bool vcf = Pc >= 0 && Pc <= Pmax_c && Pc != ghost_c && Pf >= 0 && Pf <= Pmax_f && Pf != ghost_f;
    const int i00 = (lc - co) * WF + (lf - fo), i01 = (lc - co) * WF + (lf + fo),
              i10 = (lc + co) * WF + (lf - fo), i11 = (lc + co) * WF + (lf + fo);
    const T rf = rfs[lf - fo], rc = rcs[lc - co];
    auto interp = [&](int lr) {
      const T *pl = raw + lr * NCOL;
      const T x00 = pl[i00], x01 = pl[i01], x10 = pl[i10], x11 = pl[i11];
      const T g0 = fo ? lerp_ref(x00, x01, rf) : x00;
      const T g1 = fo ? lerp_ref(x10, x11, rf) : x10;
      return co ? lerp_ref(g0, g1, rc) : g0;
    };
    auto plane_ok = [&](int lr) {
      const int P = r_lo + lr;
      return P >= 0 && P <= Pmax_r && P != ghost_r;
    };
    const bool odd_cf = (co | fo) != 0;
    T Gp = interp(0);
    cf[tid] = (vcf && odd_cf && plane_ok(0)) ? raw[tid] - Gp : (T)0;
#pragma unroll
    for (int lr = 1; lr < WR; lr += 2) {
      const T Gn = interp(lr + 1);
      cf[(lr + 1) * NCOL + tid] = (vcf && odd_cf && plane_ok(lr + 1)) ? raw[(lr + 1) * NCOL + tid] - Gn : (T)0;
      const T res = lerp_ref(Gp, Gn, rrs[lr - 1]);
      cf[lr * NCOL + tid] = (vcf && plane_ok(lr)) ? raw[lr * NCOL + tid] - res : (T)0;
      Gp = Gn;
    }
  }
  __syncthreads();

  // ---- phase 2a: owned nodes to HBM in output order: class (pr, pc, pf) = parities, then the
  // box's coarse index (a, b, c), c fastest; four per thread. Class 0 = coarse nodes (raw value,
  // no coefficient). Out-of-dictionary values are counted per wave here; their slots and their
  // list entries follow behind the next two barriers.
  constexpr int NB = TR * TC * TF;
  int64_t qd[4];
  size_t lin[4];
  bool ol[4];
  unsigned long long omask[4];
  const int wave = tid >> 6, lane = tid & 63;
  {
    unsigned nout = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int o = k * NT + tid;
      const int cls = o / NB, w = o % NB;
      const int a = w / (TC * TF), bb = (w / TF) % TC, c = w % TF;
      const int pr = (cls >> 2) & 1, pc = (cls >> 1) & 1, pf = cls & 1;
      const int wr_ = 2 * a + 2 + pr, wc_ = 2 * bb + 2 + pc, wf_ = 2 * c + 2 + pf;
      const int Pr = r_lo + wr_, Pcc = c_lo + wc_, Pff = f_lo + wf_;
      const bool valid = Pr <= Pmax_r && Pr != ghost_r && Pcc <= Pmax_c && Pcc != ghost_c &&
                         Pff <= Pmax_f && Pff != ghost_f;
      const int e = wr_ * NCOL + wc_ * WF + wf_;
      if (cls == 0 && valid) A.coarse[((size_t)(R0 + a) * mc + (C0 + bb)) * mf + (F0 + c)] = raw[e];
      lin[k] = (size_t)((pr ? mr : 0) + R0 + a) * A.dI + (size_t)((pc ? mc : 0) + C0 + bb) * A.dJ +
               (size_t)((pf ? mf : 0) + F0 + c);
      const bool on = valid && cls != 0;
      const T v = cf[e];
      ol[k] = false;
      omask[k] = 0;
      if (OUT == OUT_T) {
        if (on) A.coef[lin[k]] = v;
      } else if (kQuant) {
        qd[k] = quantize_fast(v, A.quantizer, A.volume);
        if (A.prep_huffman) {
          qd[k] += A.dict_size / 2;
          ol[k] = on && !(qd[k] >= 0 && qd[k] < A.dict_size);
        }
        omask[k] = __ballot(ol[k]);
        nout += __popcll(omask[k]);
        if (on) {
          const int64_t st = ol[k] ? 0 : qd[k];
          if (A.q16) A.q16[lin[k]] = (uint16_t)st;
          else A.q[lin[k]] = st;
        }
      }
    }
    if (kQuant && lane == 0) wcnt[wave] = nout;
  }
  // ---- phase 2b: f-sweep of every window row at the box's coarse columns -----------------------
  for (int e = tid; e < G::N1; e += NT) {
    const int jf = e % TF, row = e / TF;  // row = lr * WC + lc
    const T *x = cf + row * WF + 2 * jf;
    T w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = wfs[k * TF + jf];
    t1[e] = mass_apply(x[0], x[1], x[2], x[3], x[4], w);
  }
  __syncthreads();
  if (kQuant && tid == 0) {
    const unsigned c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
    const unsigned total = c0 + c1 + c2 + c3;
    wbase_lo[0] = 0;
    wbase_lo[1] = c0;
    wbase_lo[2] = c0 + c1;
    wbase_lo[3] = c0 + c1 + c2;
    gbase = total ? atomicAdd(A.outlier_count, (unsigned long long)total) : 0ull;
  }
  // ---- phase 3: c-sweep ----------------------------------------------------------------------
  for (int e = tid; e < G::N2; e += NT) {
    const int jf = e % TF, jc = (e / TF) % TC, lr = e / (TF * TC);
    const T *x = t1 + (lr * WC + 2 * jc) * TF + jf;
    T w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = wcs[k * TC + jc];
    t2[e] = mass_apply(x[0], x[TF], x[2 * TF], x[3 * TF], x[4 * TF], w);
  }
  __syncthreads();
  if (kQuant) {
    // outlier entries (LinearQuantization.hpp:208-241): slot = workgroup base + waves before +
    // values of this wave before
    unsigned long long at = gbase + wbase_lo[wave];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (ol[k]) {
        const unsigned long long o = at + __popcll(omask[k] & ((1ULL << lane) - 1ULL));
        if (o < A.outlier_cap) {
          A.outlier_idx[o] = lin[k];
          A.outlier_val[o] = qd[k];
        }
      }
      at += __popcll(omask[k]);
    }
  }
  // ---- phase 4: r-sweep -> load vector ---------------------------------------------------------
  for (int e = tid; e < TR * TC * TF; e += NT) {
    const int jf = e % TF, jc = (e / TF) % TC, jr = e / (TF * TC);
    if (R0 + jr < mr && C0 + jc < mc && F0 + jf < mf) {
      const T *x = t2 + ((2 * jr) * TC + jc) * TF + jf;
      constexpr int S = TC * TF;
      T w[9];
#pragma unroll
      for (int k = 0; k < 9; k++) w[k] = wrs[k * TR + jr];
      A.load[((size_t)(R0 + jr) * mc + (C0 + jc)) * mf + (F0 + jf)] =
          mass_apply(x[0], x[S], x[2 * S], x[3 * S], x[4 * S], w);
    }
  }
}

} // namespace mgh
