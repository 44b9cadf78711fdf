/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * MGARD-CPU (namespace mgard::, the serial CPU code path of BASELINE.json configs[0]) restated
 * for ONE dimension in plain C, double precision: hierarchy, decomposition / recomposition and the
 * s = infinity quantizer of mgard::compress (reference include/compress.tpp:34-63). It differs
 * from MGARD-X (oracle/mgx_oracle_impl.h) exactly where SURVEY.md section 9 says: the hierarchy
 * of a non-dyadic size (its coarser levels are index subsets of the finest grid, so they are
 * non-uniform even on a uniform grid), prolongation and restriction written with
 * (x_r - x_m) * width_reciprocal, one quantum for the whole array, x / quantum instead of
 * x * (1 / quantum).
 *
 * PINNING STATUS: pinned -- the reference's own golden vectors for this code path
 * (tests/src/test_decompose.cpp:277-338 "1D, dyadic, uniform", :433-457 "1D, dyadic, nonuniform",
 * :549-600 recomposition; tests/src/test_LinearQuantizer.cpp:94-111) are MGARD-CPU results in
 * exactly this form; tests/test_mgard_cpu_1d.py checks them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* floor(log2(n - 1)): the number of levels of the largest dyadic size 2^k + 1 <= n
 * (TensorMeshHierarchy.tpp:22-30) */
static int nlevel_from_size(uint64_t n) {
  int k = 0;
  while (((uint64_t)1 << (k + 1)) + 1 <= n) k++;
  return k;
}

/* Number of levels L and the node indices of level l inside the finest grid
 * (TensorMeshHierarchy.tpp:52-113): sizes 2, 3, 5, ..., 2^k + 1 [, n if n is not dyadic];
 * index j of a level of size m is (j (n - 1)) / (m - 1). */
int mgcpu1d_levels(uint64_t n) {
  const int k = nlevel_from_size(n);
  return ((((uint64_t)1 << k) + 1) != n) ? k + 1 : k;
}
uint64_t mgcpu1d_level_size(uint64_t n, int l) {
  const int L = mgcpu1d_levels(n);
  if (l == L) return n;
  return ((uint64_t)1 << l) + 1;
}
static void level_indices(uint64_t n, int l, uint64_t *idx) {
  const uint64_t m = mgcpu1d_level_size(n, l);
  for (uint64_t j = 0; j < m; j++) idx[j] = (j * (n - 1)) / (m - 1);
}

/* mass matrix of the level grid, in place on the level's nodes (TensorMassMatrix.tpp:15-90) */
static void mass_apply(const double *x, const uint64_t *idx, uint64_t m, double *v) {
  double x_middle = x[idx[0]], x_right = x[idx[1]];
  double h_right = x_right - x_middle, h_left;
  double v_left, v_middle = v[idx[0]], v_right = v[idx[1]];
  uint64_t out_middle = idx[0], out_right = idx[1];
  v[out_middle] = h_right / 3 * v_middle + h_right / 6 * v_right;
  for (uint64_t j = 2; j < m; j++) {
    x_middle = x_right;
    h_left = h_right;
    v_left = v_middle;
    v_middle = v_right;
    out_middle = out_right;
    x_right = x[idx[j]];
    h_right = x_right - x_middle;
    out_right = idx[j];
    v_right = v[out_right];
    v[out_middle] = h_left / 6 * v_left + (h_left + h_right) / 3 * v_middle + h_right / 6 * v_right;
  }
  h_left = h_right;
  v_left = v_middle;
  v_middle = v_right;
  out_middle = out_right;
  v[out_middle] = h_left / 6 * v_left + h_left / 3 * v_middle;
}

/* inverse of the mass matrix of the level grid: Thomas algorithm with precomputed divisors
 * (TensorMassMatrix.tpp:123-290) */
static void mass_inverse(const double *x, const uint64_t *idx, uint64_t m, double *v) {
  double *div = (double *)malloc(sizeof(double) * m);
  {
    double x_middle = x[idx[0]], x_right = x[idx[1]], h_right = x_right - x_middle, h_left;
    div[0] = 2 * h_right / 6;
    for (uint64_t j = 1; j + 1 < m; j++) {
      x_middle = x_right;
      h_left = h_right;
      x_right = x[idx[j + 1]];
      h_right = x_right - x_middle;
      const double a_j = h_left / 6;
      const double w = a_j / div[j - 1];
      div[j] = 2 * (h_left + h_right) / 6 - w * a_j;
    }
    h_left = h_right;
    const double a_j = h_left / 6;
    const double w = a_j / div[m - 2];
    div[m - 1] = 2 * h_left / 6 - w * a_j;
  }
  double x_middle = x[idx[0]], x_right = x[idx[1]], h_right = x_right - x_middle, h_left;
  double rhs_previous = v[idx[0]];
  for (uint64_t j = 1; j + 1 < m; j++) {
    x_middle = x_right;
    h_left = h_right;
    x_right = x[idx[j + 1]];
    h_right = x_right - x_middle;
    const double a_j = h_left / 6;
    const double w = a_j / div[j - 1];
    rhs_previous = v[idx[j]] -= w * rhs_previous;
  }
  {
    x_middle = x_right;
    h_left = h_right;
    const double a_j = h_left / 6;
    const double w = a_j / div[m - 2];
    v[idx[m - 1]] -= w * rhs_previous;
  }
  double x_next = v[idx[m - 1]] /= div[m - 1];
  for (uint64_t k = 2; k <= m; k++) {
    const uint64_t j = m - k;
    x_right = x_middle;
    x_middle = x[idx[j]];
    h_right = x_right - x_middle;
    const double c_j = h_right / 6;
    v[idx[j]] -= c_j * x_next;
    x_next = v[idx[j]] /= div[j];
  }
  free(div);
}

/* buffer[new nodes] += interpolant of the old (coarse) nodes (TensorProlongation.tpp:22-69) */
static void prolongation_add(const double *x, const uint64_t *fine, uint64_t mf, const uint64_t *coarse,
                             uint64_t mc, double *v) {
  uint64_t P = 1; /* fine[0] == coarse[0] */
  double x_right = x[coarse[0]], v_right = v[coarse[0]];
  for (uint64_t p = 1; p < mc; p++) {
    const double x_left = x_right, v_left = v_right;
    const uint64_t i = coarse[p];
    x_right = x[i];
    v_right = v[i];
    const double width_reciprocal = 1 / (x_right - x_left);
    uint64_t I;
    while (P < mf && (I = fine[P++]) != i) {
      const double x_middle = x[I];
      v[I] += (v_left * (x_right - x_middle) + v_right * (x_middle - x_left)) * width_reciprocal;
    }
  }
}

/* restriction onto the old nodes (TensorRestriction.tpp:23-71) */
static void restriction(const double *x, const uint64_t *fine, uint64_t mf, const uint64_t *coarse,
                        uint64_t mc, double *v) {
  uint64_t P = 1;
  double x_right = x[coarse[0]];
  uint64_t out_right = coarse[0];
  for (uint64_t p = 1; p < mc; p++) {
    const double x_left = x_right;
    const uint64_t out_left = out_right;
    const uint64_t i = coarse[p];
    x_right = x[i];
    out_right = i;
    const double width_reciprocal = 1 / (x_right - x_left);
    uint64_t I;
    while (P < mf && (I = fine[P++]) != i) {
      const double x_middle = x[I];
      const double v_middle = v[I];
      v[out_left] += v_middle * (x_right - x_middle) * width_reciprocal;
      v[out_right] += v_middle * (x_middle - x_left) * width_reciprocal;
    }
  }
}

/* mgard::decompose (decompose.tpp:129-174) in natural (unshuffled) node order: on return v holds
 * the multilevel coefficients of every node at the node's own position. x: n coordinates. */
int mgcpu1d_decompose(uint64_t n, const double *x, double *v) {
  const int L = mgcpu1d_levels(n);
  double *buffer = (double *)malloc(sizeof(double) * n);
  uint64_t *fine = (uint64_t *)malloc(sizeof(uint64_t) * n);
  uint64_t *coarse = (uint64_t *)malloc(sizeof(uint64_t) * n);
  char *is_old = (char *)malloc(n);
  for (int l = L; l > 0; l--) {
    const uint64_t mf = mgcpu1d_level_size(n, l), mc = mgcpu1d_level_size(n, l - 1);
    level_indices(n, l, fine);
    level_indices(n, l - 1, coarse);
    memset(is_old, 0, n);
    for (uint64_t j = 0; j < mc; j++) is_old[coarse[j]] = 1;
    /* copy_on_old_zero_on_new */
    for (uint64_t j = 0; j < mf; j++) buffer[fine[j]] = is_old[fine[j]] ? v[fine[j]] : 0.0;
    prolongation_add(x, fine, mf, coarse, mc, buffer);
    /* zero_on_old_subtract_and_copy_back_on_new */
    for (uint64_t j = 0; j < mf; j++) {
      const uint64_t i = fine[j];
      if (is_old[i]) buffer[i] = 0.0;
      else buffer[i] = (v[i] -= buffer[i]);
    }
    mass_apply(x, fine, mf, buffer);
    restriction(x, fine, mf, coarse, mc, buffer);
    mass_inverse(x, coarse, mc, buffer);
    /* add_on_old_add_on_new (level l - 1: every node of the coarse level) */
    for (uint64_t j = 0; j < mc; j++) v[coarse[j]] += buffer[coarse[j]];
  }
  free(buffer);
  free(fine);
  free(coarse);
  free(is_old);
  return L;
}

/* mgard::recompose (decompose.tpp:176-226): the inverse, level 1 upwards */
int mgcpu1d_recompose(uint64_t n, const double *x, double *v) {
  const int L = mgcpu1d_levels(n);
  double *buffer = (double *)malloc(sizeof(double) * n);
  uint64_t *fine = (uint64_t *)malloc(sizeof(uint64_t) * n);
  uint64_t *coarse = (uint64_t *)malloc(sizeof(uint64_t) * n);
  char *is_old = (char *)malloc(n);
  for (int l = 1; l <= L; l++) {
    const uint64_t mf = mgcpu1d_level_size(n, l), mc = mgcpu1d_level_size(n, l - 1);
    level_indices(n, l, fine);
    level_indices(n, l - 1, coarse);
    memset(is_old, 0, n);
    for (uint64_t j = 0; j < mc; j++) is_old[coarse[j]] = 1;
    /* correction from the coefficients of the new nodes */
    for (uint64_t j = 0; j < mf; j++) buffer[fine[j]] = is_old[fine[j]] ? 0.0 : v[fine[j]];
    mass_apply(x, fine, mf, buffer);
    restriction(x, fine, mf, coarse, mc, buffer);
    mass_inverse(x, coarse, mc, buffer);
    for (uint64_t j = 0; j < mc; j++) v[coarse[j]] -= buffer[coarse[j]];
    /* interpolate the corrected coarse values and add them to the coefficients */
    for (uint64_t j = 0; j < mf; j++) buffer[fine[j]] = is_old[fine[j]] ? v[fine[j]] : 0.0;
    prolongation_add(x, fine, mf, coarse, mc, buffer);
    for (uint64_t j = 0; j < mf; j++)
      if (!is_old[fine[j]]) v[fine[j]] += buffer[fine[j]];
  }
  free(buffer);
  free(fine);
  free(coarse);
  free(is_old);
  return L;
}

/* s = infinity: one quantum 2 tol / ((L + 1)(1 + 3^d)), d = 1
 * (TensorMultilevelCoefficientQuantizer.tpp:13-26); n = copysign(0.5 + |x / quantum|, x)
 * truncated (LinearQuantizer.tpp:20-26), dequantize = quantum * n (:41-46) */
double mgcpu1d_quantum(uint64_t n, double tol) {
  const int L = mgcpu1d_levels(n);
  return (2 * tol) / ((L + 1) * (1 + pow(3, 1)));
}
void mgcpu1d_quantize(uint64_t n, const double *v, double quantum, int64_t *q) {
  for (uint64_t i = 0; i < n; i++) q[i] = (int64_t)copysign(0.5 + fabs(v[i] / quantum), v[i]);
}
void mgcpu1d_dequantize(uint64_t n, const int64_t *q, double quantum, double *v) {
  for (uint64_t i = 0; i < n; i++) v[i] = quantum * (double)q[i];
}
