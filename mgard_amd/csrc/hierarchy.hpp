// Host-side level hierarchy of the MI355X-native MGARD-X hot path.
//
// Builds, once per shape, everything the kernels need that depends only on the
// grid: level shapes (n -> n/2+1), per-level node spacings, interpolation
// ratios, the per-coarse-node constants of the fused mass-matrix/restriction
// stencil and the Thomas factors. All of it is computed in the working type T
// with the reference's operation order and no FMA contraction, so the values
// are bit-identical to what mgard_x::Hierarchy holds
// (reference: include/mgard-x/Hierarchy/Hierarchy.hpp:23-190, 193-418, 689-708).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace mgh {

constexpr int kMaxDim = 5;

template <typename T> struct DimLevel {
  uint64_t n = 1;        // nodes of this dim on this level
  std::vector<T> dist;   // n entries      (Hierarchy.hpp:23-50, 82-109)
  std::vector<T> ratio;  // n entries      (Hierarchy.hpp:53-79)
  std::vector<T> am, bm; // n+1 entries    (Hierarchy.hpp:112-162)
};

// Constants of mass_trans (Correction/LPKFunctor.h:77-93) for coarse node j of a
// dim with fine spacing h1..h4 = dist[2j-2 .. 2j+1] (0 outside the grid):
//   tb = a*c[0] + b*c[1] + c*c[2]; tc = b*c[2] + c*c[3] + d*c[4];
//   td = c*c[4] + d*c[5] + e*c[6]; out = tc + (tb*c[7] + td*c[8])
// c = {h1/6, (h1+h2)/3, h2/6, (h2+h3)/3, h3/6, (h3+h4)/3, h4/6, r1, r4}.
constexpr int kMassCoef = 9;

template <typename T> struct HostHierarchy {
  int D = 0;
  int L = 0; // l_target
  bool uniform = true;
  bool normalize_coordinates = true; // Config::normalize_coordinates (norm scaling)
  uint64_t shape[kMaxDim] = {1, 1, 1, 1, 1};
  std::vector<std::vector<uint64_t>> level_shape;  // [l][d]
  std::vector<std::vector<DimLevel<T>>> lv;        // [l][d]
  std::vector<std::vector<T>> coords;              // [d]
  std::vector<std::vector<int>> marks;             // [d][i], Hierarchy.hpp:261-281
  std::vector<std::vector<T>> vol, vol_recip;      // [l][d], Hierarchy.hpp:165-190

  // returns false on an invalid shape (any dim < 3; Hierarchy.hpp:742-756)
  bool init(int D_, const uint64_t *shape_, const T *const *coords_, bool normalize,
            uint64_t max_level) {
    D = D_;
    if (D < 1 || D > kMaxDim) return false;
    for (int d = 0; d < D; d++) {
      if (shape_[d] < 3) return false;
      shape[d] = shape_[d];
    }
    uniform = (coords_ == nullptr);
    std::vector<std::vector<uint64_t>> seq(D);
    size_t nlevel = std::numeric_limits<size_t>::max();
    for (int d = 0; d < D; d++) {
      uint64_t n = shape[d];
      while (n > 2) {
        seq[d].push_back(n);
        n = n / 2 + 1;
      }
      seq[d].push_back(2);
      nlevel = std::min(nlevel, seq[d].size());
    }
    uint64_t lt = nlevel - 1;
    if (max_level < lt) lt = max_level;
    L = (int)lt;
    level_shape.assign(L + 1, std::vector<uint64_t>(D));
    for (int l = 0; l <= L; l++)
      for (int d = 0; d < D; d++) level_shape[l][d] = seq[d][L - l];

    coords.assign(D, {});
    marks.assign(D, {});
    lv.assign(L + 1, std::vector<DimLevel<T>>(D));
    vol.assign(L + 1, std::vector<T>(D));
    vol_recip.assign(L + 1, std::vector<T>(D));
    for (int d = 0; d < D; d++) {
      const uint64_t n = shape[d];
      coords[d].resize(n);
      for (uint64_t i = 0; i < n; i++) {
        if (coords_)
          coords[d][i] = coords_[d][i];
        else
          coords[d][i] = normalize ? (T)i / (T)(n - 1) : (T)i; // Hierarchy.hpp:695-703
      }
      marks[d].resize(n);
      {
        uint64_t i = 0;
        for (int l = 0; l <= L; l++)
          for (; i < level_shape[l][d]; i++) marks[d][i] = l;
      }
      for (int l = L; l >= 0; l--) {
        DimLevel<T> &q = lv[l][d];
        q.n = level_shape[l][d];
        q.dist.assign(q.n, 0);
        q.ratio.assign(q.n, 0);
        if (l == L) {
          for (uint64_t i = 0; i + 1 < q.n; i++) q.dist[i] = coords[d][i + 1] - coords[d][i];
        } else {
          const DimLevel<T> &f = lv[l + 1][d];
          for (uint64_t i = 0; i + 1 < q.n; i++) q.dist[i] = f.dist[2 * i] + f.dist[2 * i + 1];
        }
        if (q.n != 2 && q.n % 2 == 0) { // split the last cell: ghost node
          T last = q.dist[q.n - 2];
          q.dist[q.n - 2] = (T)(last / 2.0);
          q.dist[q.n - 1] = (T)(last / 2.0);
        }
        for (uint64_t i = 0; i + 2 < q.n; i++) q.ratio[i] = q.dist[i] / (q.dist[i + 1] + q.dist[i]);
        if (q.n % 2 == 0) q.ratio[q.n - 2] = q.dist[q.n - 2] / (q.dist[q.n - 1] + q.dist[q.n - 2]);
      }
      for (int l = 0; l <= L; l++) {
        DimLevel<T> &q = lv[l][d];
        const uint64_t m = q.n;
        vol[l][d] = (T)(1.0 / (T)(m - 1));
        vol_recip[l][d] = (T)(1.0 / vol[l][d]);
        std::vector<T> ha(m + 1, 0), hb(m + 1, 0);
        hb[0] = 2 * q.dist[0] / 6;
        for (uint64_t i = 1; i + 1 < m; i++) {
          T a_j = q.dist[i - 1] / 6;
          T w = a_j / hb[i - 1];
          hb[i] = 2 * (q.dist[i - 1] + q.dist[i]) / 6 - w * a_j;
          ha[i] = a_j;
        }
        {
          T a_j = q.dist[m - 2] / 6;
          T w = a_j / hb[m - 2];
          hb[m - 1] = 2 * q.dist[m - 2] / 6 - w * a_j;
          ha[m - 1] = a_j;
        }
        q.am.assign(m + 1, 0);
        q.bm.assign(m + 1, 0);
        for (uint64_t i = 0; i < m; i++) q.am[i] = ha[i];
        q.bm[0] = 1;
        for (uint64_t i = 0; i < m; i++) q.bm[i + 1] = hb[i];
      }
    }
    return true;
  }

  uint64_t total() const {
    uint64_t t = 1;
    for (int d = 0; d < D; d++) t *= shape[d];
    return t;
  }

  // mass-trans constants for restricting dim d from level l to l-1; SoA:
  // out[k * nc + j], k in [0, kMassCoef)
  std::vector<T> mass_table(int l, int d) const {
    const DimLevel<T> &f = lv[l][d];
    const uint64_t n = f.n, nc = lv[l - 1][d].n;
    std::vector<T> t(kMassCoef * nc);
    for (uint64_t j = 0; j < nc; j++) {
      T h1 = (j >= 1) ? f.dist[2 * j - 2] : (T)0;
      T h2 = (j >= 1) ? f.dist[2 * j - 1] : (T)0;
      T h3 = (2 * j < n) ? f.dist[2 * j] : (T)0;
      T h4 = (2 * j + 1 < n) ? f.dist[2 * j + 1] : (T)0;
      T r1 = (h1 + h2 != 0) ? h1 / (h1 + h2) : (T)0;
      T r4 = (h3 + h4 != 0) ? h4 / (h3 + h4) : (T)0;
      T c[kMassCoef] = {h1 / 6, (h1 + h2) / 3, h2 / 6, (h2 + h3) / 3, h3 / 6,
                        (h3 + h4) / 3, h4 / 6, r1, r4};
      for (int k = 0; k < kMassCoef; k++) t[k * nc + j] = c[k];
    }
    return t;
  }

  // Thomas tables of level l, dim d (n = nodes): SoA out[k * n + i]:
  //  k=0 forward multiplier am[i]/bm[i]      (x[i] -= x[i-1] * f[i], IPKFunctor.h:127)
  //  k=1 backward am[i+1], k=2 backward bm[i+1]
  //      (x[i] = (x[i] - am[i+1]*x[i+1]) / bm[i+1], IPKFunctor.h:147)
  std::vector<T> thomas_table(int l, int d) const {
    const DimLevel<T> &q = lv[l][d];
    const uint64_t n = q.n;
    std::vector<T> t(3 * n);
    for (uint64_t i = 0; i < n; i++) {
      t[i] = q.am[i] / q.bm[i];
      t[n + i] = q.am[i + 1];
      t[2 * n + i] = q.bm[i + 1];
    }
    return t;
  }

  // Denominators of the level quantizers: quantizer[l] = (T)(abs_tol / den[l])
  // (Quantization/LinearQuantization.hpp:495-545, MultiDim decomposition)
  void quantizer_denominators(T s, double *den) const {
    const uint64_t l_target = (uint64_t)L;
    const uint64_t dof = total();
    for (int l = 0; l <= L; l++) {
      if (s == std::numeric_limits<T>::infinity())
        den[l] = ((l_target + 1) * (1 + std::pow(3, D)));
      else
        den[l] = (std::exp2(s * l) * std::sqrt((double)dof));
    }
  }

  void quantizers(int ebtype_rel0_abs1, T tol, T s, T norm, bool reciprocal, T *out) const {
    double abs_tol = tol;
    if (ebtype_rel0_abs1 == 0) abs_tol *= norm;
    abs_tol *= 2;
    std::vector<double> den(L + 1);
    quantizer_denominators(s, den.data());
    for (int l = 0; l <= L; l++) {
      out[l] = (abs_tol) / den[l];
      if (reciprocal) out[l] = 1.0f / out[l];
    }
  }

  // sqrt(prod_d level_volumes[level][d]) in the kernel's order d = D-1..0
  // (LinearQuantization.hpp:186-195)
  T level_volume(int level, bool reciprocal) const {
    T v = 1;
    for (int d = D - 1; d >= 0; d--) v *= reciprocal ? vol_recip[level][d] : vol[level][d];
    return std::sqrt(v);
  }
};

} // namespace mgh
