/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle (included twice by mgx_oracle.c with
 * REAL = float / double).  It is a plain-loop restatement of the MGARD-X
 * (namespace mgard_x) multilevel decomposition + level-wise linear quantizer,
 * following the reference's *non-FMA* expression order (MGARD_X_FMA is never
 * defined by the reference build).  Every function cites the reference
 * file:line (relative to the reference checkout) it restates.
 *
 * Compile with -ffp-contract=off.  OpenMP pragmas only parallelise over
 * independent pencils/elements, so results do not depend on thread count.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef struct FN(mgxo_hier) {
  int D;  /* user dims, 1..MGXO_MAXD */
  int L;  /* l_target */
  int uniform;
  uint64_t shape[MGXO_MAXD];
  uint64_t lshape[MGXO_MAXL + 1][MGXO_MAXD];
  REAL *coords[MGXO_MAXD];
  REAL *dist[MGXO_MAXL + 1][MGXO_MAXD];
  REAL *ratio[MGXO_MAXL + 1][MGXO_MAXD];
  REAL *am[MGXO_MAXL + 1][MGXO_MAXD]; /* n+1 entries */
  REAL *bm[MGXO_MAXL + 1][MGXO_MAXD]; /* n+1 entries */
  REAL vol[MGXO_MAXL + 1][MGXO_MAXD];       /* level_volumes (uniform per level/dim) */
  REAL vol_recip[MGXO_MAXL + 1][MGXO_MAXD]; /* level_volumes_reciprocal */
  int *marks[MGXO_MAXD];
} FN(mgxo_hier);

/* Hierarchy.hpp:23-50 (coord_to_dist) */
static void FN(coord_to_dist)(uint64_t n, const REAL *coord, REAL *dist) {
  for (uint64_t i = 0; i < n; i++) dist[i] = 0;
  if (n <= 1) return;
  for (uint64_t i = 0; i + 1 < n; i++) dist[i] = coord[i + 1] - coord[i];
  if (n != 2 && n % 2 == 0) {
    REAL last = dist[n - 2];
    dist[n - 2] = (REAL)(last / 2.0);
    dist[n - 1] = (REAL)(last / 2.0);
  }
}

/* Hierarchy.hpp:53-79 (dist_to_ratio) */
static void FN(dist_to_ratio)(uint64_t n, const REAL *dist, REAL *ratio) {
  for (uint64_t i = 0; i < n; i++) ratio[i] = 0;
  if (n <= 1) return;
  for (uint64_t i = 0; i + 2 < n; i++) ratio[i] = dist[i] / (dist[i + 1] + dist[i]);
  if (n % 2 == 0) ratio[n - 2] = dist[n - 2] / (dist[n - 1] + dist[n - 2]);
}

/* Hierarchy.hpp:82-109 (reduce_dist) */
static void FN(reduce_dist)(uint64_t n, const REAL *dist, REAL *dist2) {
  uint64_t n2 = n / 2 + 1;
  for (uint64_t i = 0; i < n2; i++) dist2[i] = 0;
  if (n <= 1) return;
  for (uint64_t i = 0; i + 1 < n2; i++) dist2[i] = dist[2 * i] + dist[2 * i + 1];
  if (n2 != 2 && n2 % 2 == 0) {
    REAL last = dist2[n2 - 2];
    dist2[n2 - 2] = (REAL)(last / 2.0);
    dist2[n2 - 1] = (REAL)(last / 2.0);
  }
}

/* Hierarchy.hpp:112-162 (calc_am_bm): device layout am[0..n)=ha, am[n]=0,
 * bm[0]=1, bm[1..n]=hb */
static void FN(calc_am_bm)(uint64_t n, const REAL *dist, REAL *am, REAL *bm) {
  REAL *ha = (REAL *)calloc(n + 1, sizeof(REAL));
  REAL *hb = (REAL *)calloc(n + 1, sizeof(REAL));
  hb[0] = 2 * dist[0] / 6;
  ha[0] = 0;
  for (uint64_t i = 1; i + 1 < n; i++) {
    REAL a_j = dist[i - 1] / 6;
    REAL w = a_j / hb[i - 1];
    hb[i] = 2 * (dist[i - 1] + dist[i]) / 6 - w * a_j;
    ha[i] = a_j;
  }
  {
    REAL a_j = dist[n - 2] / 6;
    REAL w = a_j / hb[n - 2];
    hb[n - 1] = 2 * dist[n - 2] / 6 - w * a_j;
    ha[n - 1] = a_j;
  }
  for (uint64_t i = 0; i < n; i++) am[i] = ha[i];
  am[n] = 0;
  bm[0] = 1;
  for (uint64_t i = 0; i < n; i++) bm[i + 1] = hb[i];
  free(ha);
  free(hb);
}

void FN(mgxo_hier_destroy)(FN(mgxo_hier) * h) {
  if (!h) return;
  for (int d = 0; d < MGXO_MAXD; d++) {
    free(h->coords[d]);
    free(h->marks[d]);
    for (int l = 0; l <= MGXO_MAXL; l++) {
      free(h->dist[l][d]);
      free(h->ratio[l][d]);
      free(h->am[l][d]);
      free(h->bm[l][d]);
    }
  }
  free(h);
}

/* Hierarchy.hpp:193-418 (init), :689-708 (create_uniform_coords).
 * coords == NULL -> uniform grid. Returns NULL on invalid shape
 * (check_shape: every dim >= 3, Hierarchy.hpp:742-756). */
FN(mgxo_hier) * FN(mgxo_hier_create)(int D, const uint64_t *shape, const REAL *const *coords,
                                     int normalize_coordinates, uint64_t max_level) {
  if (D < 1 || D > MGXO_MAXD) return NULL;
  for (int d = 0; d < D; d++)
    if (shape[d] < 3) return NULL;
  FN(mgxo_hier) *h = (FN(mgxo_hier) *)calloc(1, sizeof(*h));
  h->D = D;
  h->uniform = (coords == NULL);
  uint64_t seq[MGXO_MAXD][MGXO_MAXL + 2];
  int nseq[MGXO_MAXD];
  int nlevel = 1 << 30;
  for (int d = 0; d < D; d++) {
    h->shape[d] = shape[d];
    uint64_t n = shape[d];
    int k = 0;
    while (n > 2) {
      seq[d][k++] = n;
      n = n / 2 + 1;
    }
    seq[d][k++] = 2;
    nseq[d] = k;
    if (k < nlevel) nlevel = k;
  }
  uint64_t L = (uint64_t)(nlevel - 1);
  if (max_level < L) L = max_level;
  h->L = (int)L;
  for (int l = 0; l <= h->L; l++)
    for (int d = 0; d < D; d++) h->lshape[l][d] = seq[d][h->L - l];
  (void)nseq;

  for (int d = 0; d < D; d++) {
    uint64_t n = shape[d];
    h->coords[d] = (REAL *)malloc(n * sizeof(REAL));
    if (coords) {
      memcpy(h->coords[d], coords[d], n * sizeof(REAL));
    } else {
      for (uint64_t i = 0; i < n; i++)
        h->coords[d][i] = normalize_coordinates ? (REAL)i / (REAL)(n - 1) : (REAL)i;
    }
    /* level_marks, Hierarchy.hpp:261-281 */
    h->marks[d] = (int *)malloc(n * sizeof(int));
    {
      uint64_t i = 0;
      for (int l = 0; l <= h->L; l++)
        for (; i < h->lshape[l][d]; i++) h->marks[d][i] = l;
    }
    /* dist / ratio, Hierarchy.hpp:291-325 */
    for (int l = h->L; l >= 0; l--) {
      uint64_t nl = h->lshape[l][d];
      h->dist[l][d] = (REAL *)malloc(nl * sizeof(REAL));
      h->ratio[l][d] = (REAL *)malloc(nl * sizeof(REAL));
      if (l == h->L)
        FN(coord_to_dist)(nl, h->coords[d], h->dist[l][d]);
      else
        FN(reduce_dist)(h->lshape[l + 1][d], h->dist[l + 1][d], h->dist[l][d]);
      FN(dist_to_ratio)(nl, h->dist[l][d], h->ratio[l][d]);
    }
    for (int l = 0; l <= h->L; l++) {
      uint64_t nl = h->lshape[l][d];
      /* calc_volume, Hierarchy.hpp:165-190 */
      h->vol[l][d] = (REAL)(1.0 / (REAL)(nl - 1));
      h->vol_recip[l][d] = (REAL)(1.0 / h->vol[l][d]);
      h->am[l][d] = (REAL *)calloc(nl + 1, sizeof(REAL));
      h->bm[l][d] = (REAL *)calloc(nl + 1, sizeof(REAL));
      FN(calc_am_bm)(nl, h->dist[l][d], h->am[l][d], h->bm[l][d]);
    }
  }
  return h;
}

int FN(mgxo_l_target)(const FN(mgxo_hier) * h) { return h->L; }
void FN(mgxo_level_shape)(const FN(mgxo_hier) * h, int l, uint64_t *out) {
  for (int d = 0; d < h->D; d++) out[d] = h->lshape[l][d];
}
const REAL *FN(mgxo_dist)(const FN(mgxo_hier) * h, int l, int d) { return h->dist[l][d]; }
const REAL *FN(mgxo_ratio)(const FN(mgxo_hier) * h, int l, int d) { return h->ratio[l][d]; }
const REAL *FN(mgxo_am)(const FN(mgxo_hier) * h, int l, int d) { return h->am[l][d]; }
const REAL *FN(mgxo_bm)(const FN(mgxo_hier) * h, int l, int d) { return h->bm[l][d]; }
const int *FN(mgxo_marks)(const FN(mgxo_hier) * h, int d) { return h->marks[d]; }

/* Coefficient/GPKFunctor.h:21-23 (non-FMA branch) */
static inline REAL FN(lerp)(REAL v0, REAL v1, REAL t) {
  REAL r = v0 + v0 * t * -1;
  r = r + t * v1;
  return r;
}

/* Correction/LPKFunctor.h:77-93 (non-FMA branch; the ratio arguments the
 * reference passes in are overwritten there, so they are not parameters here) */
static inline REAL FN(mass_trans)(REAL a, REAL b, REAL c, REAL d, REAL e, REAL h1, REAL h2,
                                  REAL h3, REAL h4) {
  REAL r1, r4, tb, tc, td;
  if (h1 + h2 != 0)
    r1 = h1 / (h1 + h2);
  else
    r1 = 0;
  if (h3 + h4 != 0)
    r4 = h4 / (h3 + h4);
  else
    r4 = 0;
  tb = a * (h1 / 6) + b * ((h1 + h2) / 3) + c * (h2 / 6);
  tc = b * (h2 / 6) + c * ((h2 + h3) / 3) + d * (h3 / 6);
  td = c * (h3 / 6) + d * ((h3 + h4) / 3) + e * (h4 / 6);
  tc += tb * r1 + td * r4;
  return tc;
}

/* Views of a D<=3 problem as (r,c,f) with leading 1s, like
 * SubArray::project(D-3,D-2,D-1) + Hierarchy::level_shape(l, dim>=D) == 1
 * (Hierarchy.hpp:569-578). */
typedef struct {
  uint64_t n[3], nc[3];          /* fine / coarse sizes */
  const REAL *ratio[3], *dist[3]; /* fine level */
  const REAL *am[3], *bm[3];      /* coarse level */
  int active[3];
} FN(lvl3);

static void FN(get_lvl3)(const FN(mgxo_hier) * h, int l, FN(lvl3) * q) {
  for (int k = 0; k < 3; k++) {
    int d = h->D - 3 + k;
    if (d < 0) {
      q->n[k] = q->nc[k] = 1;
      q->ratio[k] = q->dist[k] = q->am[k] = q->bm[k] = NULL;
      q->active[k] = 0;
    } else {
      q->n[k] = h->lshape[l][d];
      q->nc[k] = h->lshape[l - 1][d];
      q->ratio[k] = h->ratio[l][d];
      q->dist[k] = h->dist[l][d];
      q->am[k] = h->am[l - 1][d];
      q->bm[k] = h->bm[l - 1][d];
      q->active[k] = 1;
    }
  }
}

/* Position of reordered index i along a dim of fine size n / coarse size nc:
 * i < nc : coarse node at fine index min(2i, n-1)  (even n: last coarse node is
 *          the real last node, GridProcessingKernel3D.hpp:204-231)
 * i >= nc: coefficient at odd fine index 2(i-nc)+1. */
#define IDX3(i, j, k, s1, s2) (((uint64_t)(i) * (s1) + (uint64_t)(j)) * (s2) + (uint64_t)(k))

/* Coefficient/GridProcessingKernel3D.hpp:21-1179 (GpkReo3D): w = natural-order
 * fine box (n[0],n[1],n[2]) compact; v receives [coarse | coefficients] in
 * reordered layout with strides (ldv1, ldv2). Interpolation order f, then c,
 * then r (:614-617, :737-744, :854-871). */
static void FN(gpk_reo)(const FN(lvl3) * q, const REAL *w, REAL *v, uint64_t ldv1,
                        uint64_t ldv2) {
  const uint64_t nr = q->n[0], nc = q->n[1], nf = q->n[2];
  const uint64_t rr = q->nc[0], cc = q->nc[1], ff = q->nc[2];
#pragma omp parallel for collapse(2) schedule(static)
  for (uint64_t i = 0; i < nr; i++) {
    for (uint64_t j = 0; j < nc; j++) {
      int ro = i >= rr, co = j >= cc;
      uint64_t rp = ro ? 2 * (i - rr) + 1 : (2 * i < nr - 1 ? 2 * i : nr - 1);
      uint64_t cp = co ? 2 * (j - cc) + 1 : (2 * j < nc - 1 ? 2 * j : nc - 1);
      for (uint64_t k = 0; k < nf; k++) {
        int fo = k >= ff;
        uint64_t fp = fo ? 2 * (k - ff) + 1 : (2 * k < nf - 1 ? 2 * k : nf - 1);
        REAL center = w[IDX3(rp, cp, fp, nc, nf)];
        if (!ro && !co && !fo) {
          v[IDX3(i, j, k, ldv1, ldv2)] = center;
          continue;
        }
        /* corner fine indices */
        uint64_t rs[2], cs[2], fs[2];
        int nrs = ro ? 2 : 1, ncs = co ? 2 : 1, nfs = fo ? 2 : 1;
        rs[0] = ro ? rp - 1 : rp; rs[1] = rp + 1;
        cs[0] = co ? cp - 1 : cp; cs[1] = cp + 1;
        fs[0] = fo ? fp - 1 : fp; fs[1] = fp + 1;
        REAL hr[2];
        for (int a = 0; a < nrs; a++) {
          REAL gc[2];
          for (int b = 0; b < ncs; b++) {
            const REAL *row = w + IDX3(rs[a], cs[b], 0, nc, nf);
            gc[b] = (nfs == 2) ? FN(lerp)(row[fs[0]], row[fs[1]], q->ratio[2][fs[0]]) : row[fs[0]];
          }
          hr[a] = (ncs == 2) ? FN(lerp)(gc[0], gc[1], q->ratio[1][cs[0]]) : gc[0];
        }
        REAL res = (nrs == 2) ? FN(lerp)(hr[0], hr[1], q->ratio[0][rs[0]]) : hr[0];
        v[IDX3(i, j, k, ldv1, ldv2)] = center - res;
      }
    }
  }
}

/* Coefficient/GridProcessingKernel3D.hpp:1231-2352 (GpkRev3D): inverse of the
 * above; v = reordered [coarse|coeff] (strides ldv1, ldv2), w = natural fine
 * box compact. */
static void FN(gpk_rev)(const FN(lvl3) * q, const REAL *v, uint64_t ldv1, uint64_t ldv2,
                        REAL *w) {
  const uint64_t nr = q->n[0], nc = q->n[1], nf = q->n[2];
  const uint64_t rr = q->nc[0], cc = q->nc[1], ff = q->nc[2];
  /* pass 1: scatter coarse nodes to their fine positions */
#pragma omp parallel for collapse(2) schedule(static)
  for (uint64_t i = 0; i < rr; i++)
    for (uint64_t j = 0; j < cc; j++) {
      uint64_t rp = 2 * i < nr - 1 ? 2 * i : nr - 1;
      uint64_t cp = 2 * j < nc - 1 ? 2 * j : nc - 1;
      for (uint64_t k = 0; k < ff; k++) {
        uint64_t fp = 2 * k < nf - 1 ? 2 * k : nf - 1;
        w[IDX3(rp, cp, fp, nc, nf)] = v[IDX3(i, j, k, ldv1, ldv2)];
      }
    }
  /* pass 2: coefficient nodes = coeff + interpolation of coarse nodes only */
#pragma omp parallel for collapse(2) schedule(static)
  for (uint64_t i = 0; i < nr; i++) {
    for (uint64_t j = 0; j < nc; j++) {
      int ro = i >= rr, co = j >= cc;
      uint64_t rp = ro ? 2 * (i - rr) + 1 : (2 * i < nr - 1 ? 2 * i : nr - 1);
      uint64_t cp = co ? 2 * (j - cc) + 1 : (2 * j < nc - 1 ? 2 * j : nc - 1);
      for (uint64_t k = 0; k < nf; k++) {
        int fo = k >= ff;
        if (!ro && !co && !fo) continue;
        uint64_t fp = fo ? 2 * (k - ff) + 1 : (2 * k < nf - 1 ? 2 * k : nf - 1);
        uint64_t rs[2], cs[2], fs[2];
        int nrs = ro ? 2 : 1, ncs = co ? 2 : 1, nfs = fo ? 2 : 1;
        rs[0] = ro ? rp - 1 : rp; rs[1] = rp + 1;
        cs[0] = co ? cp - 1 : cp; cs[1] = cp + 1;
        fs[0] = fo ? fp - 1 : fp; fs[1] = fp + 1;
        REAL hr[2];
        for (int a = 0; a < nrs; a++) {
          REAL gc[2];
          for (int b = 0; b < ncs; b++) {
            const REAL *row = w + IDX3(rs[a], cs[b], 0, nc, nf);
            gc[b] = (nfs == 2) ? FN(lerp)(row[fs[0]], row[fs[1]], q->ratio[2][fs[0]]) : row[fs[0]];
          }
          hr[a] = (ncs == 2) ? FN(lerp)(gc[0], gc[1], q->ratio[1][cs[0]]) : gc[0];
        }
        REAL res = v[IDX3(i, j, k, ldv1, ldv2)];
        res += (nrs == 2) ? FN(lerp)(hr[0], hr[1], q->ratio[0][rs[0]]) : hr[0];
        w[IDX3(rp, cp, fp, nc, nf)] = res;
      }
    }
  }
}

/* One fused mass-matrix + restriction line, Correction/LinearProcessingKernel3D.hpp
 * :27-400 (Lpk1Reo3D), :449-717 (Lpk2), :762-1048 (Lpk3).
 * even(j) for j in [0,ncoarse) else 0; odd(j) for j in [0,n-ncoarse) else 0
 * (this drops the ghost coefficient of an even-sized dim, :52,177-203);
 * h = dist[2j-2..2j+1], 0 out of [0,n). stride s between consecutive line
 * elements of in (even part at in_e, odd part at in_o) and so for out. */
static inline void FN(mass_trans_line)(uint64_t n, uint64_t ncoarse, const REAL *dist,
                                       const REAL *in_e, int e_zero, const REAL *in_o,
                                       uint64_t si, REAL *out, uint64_t so) {
  const uint64_t nodd = n - ncoarse;
  for (uint64_t j = 0; j < ncoarse; j++) {
    REAL a = (j >= 1 && !e_zero) ? in_e[(j - 1) * si] : 0;
    REAL b = (j >= 1 && j - 1 < nodd) ? in_o[(j - 1) * si] : 0;
    REAL c = e_zero ? 0 : in_e[j * si];
    REAL d = (j < nodd) ? in_o[j * si] : 0;
    REAL e = (j + 1 < ncoarse && !e_zero) ? in_e[(j + 1) * si] : 0;
    REAL h1 = (j >= 1) ? dist[2 * j - 2] : 0;
    REAL h2 = (j >= 1) ? dist[2 * j - 1] : 0;
    REAL h3 = (2 * j < n) ? dist[2 * j] : 0;
    REAL h4 = (2 * j + 1 < n) ? dist[2 * j + 1] : 0;
    out[j * so] = FN(mass_trans)(a, b, c, d, e, h1, h2, h3, h4);
  }
}

/* Thomas solve on one line, Correction/IPKFunctor.h:111-149 (non-FMA) with the
 * am/bm indexing of IterativeProcessingKernel3D.hpp:108-124 (forward) and
 * :223-262 (backward, reversed am/bm). am, bm have n+1 entries. */
static inline void FN(thomas_line)(uint64_t n, const REAL *am, const REAL *bm, REAL *x,
                                   uint64_t s) {
  REAL prev = 0;
  for (uint64_t i = 0; i < n; i++) {
    REAL cur = x[i * s];
    cur = cur - prev * (am[i] / bm[i]);
    x[i * s] = cur;
    prev = cur;
  }
  prev = 0;
  for (uint64_t k = 0; k < n; k++) {
    uint64_t i = n - 1 - k;
    REAL cur = x[i * s];
    cur = (cur - am[n - k] * prev) / bm[n - k];
    x[i * s] = cur;
    prev = cur;
  }
}

/* ---- test hooks: the three line operators of the decomposition on their own, so that the
 * operator vectors the reference's tests hold (tests/src/test_TensorMassMatrix.cpp,
 * test_TensorRestriction.cpp, test_TensorProlongation.cpp: known answers on custom spacing) can
 * pin the expressions above -- tests/test_oracle_operator_goldens.py. `fine`: the level-l nodes
 * of dimension d in natural order; odd sizes only (the ghost node of an even size has no
 * counterpart in MGARD-CPU, whose vectors these are). Return 0, or -1 on a bad argument. ---- */
int FN(mgxo_op_lerp)(const FN(mgxo_hier) * h, int l, int d, const REAL *fine, REAL *out_odd) {
  if (l < 1 || l > h->L || d < 0 || d >= h->D) return -1;
  const uint64_t n = h->lshape[l][d];
  if (n % 2 == 0) return -1;
  for (uint64_t p = 1; p + 1 < n; p += 2)  /* GridProcessingKernel3D.hpp:614-617 */
    out_odd[p / 2] = FN(lerp)(fine[p - 1], fine[p + 1], h->ratio[l][d][p - 1]);
  return 0;
}
int FN(mgxo_op_mass_trans)(const FN(mgxo_hier) * h, int l, int d, const REAL *fine, REAL *coarse_out) {
  if (l < 1 || l > h->L || d < 0 || d >= h->D) return -1;
  const uint64_t n = h->lshape[l][d], nc = h->lshape[l - 1][d];
  if (n % 2 == 0) return -1;
  REAL *e = (REAL *)malloc(nc * sizeof(REAL)), *o = (REAL *)malloc((n - nc + 1) * sizeof(REAL));
  for (uint64_t j = 0; j < nc; j++) e[j] = fine[2 * j];
  for (uint64_t j = 0; j < n - nc; j++) o[j] = fine[2 * j + 1];
  FN(mass_trans_line)(n, nc, h->dist[l][d], e, 0, o, 1, coarse_out, 1);
  free(e);
  free(o);
  return 0;
}
int FN(mgxo_op_thomas)(const FN(mgxo_hier) * h, int l, int d, REAL *x) {
  if (l < 0 || l > h->L || d < 0 || d >= h->D) return -1;
  FN(thomas_line)(h->lshape[l][d], h->am[l][d], h->bm[l][d], x, 1);
  return 0;
}

/* Correction/CalcCorrection3D.hpp:26-185: LPK1,2,3 then IPK1,2,3 on the
 * reordered coefficient array v (strides ldv1, ldv2); returns a freshly
 * allocated compact (rr,cc,ff) correction. */
static REAL *FN(calc_correction)(const FN(lvl3) * q, const REAL *v, uint64_t ldv1,
                                 uint64_t ldv2) {
  const uint64_t nr = q->n[0], nc = q->n[1], nf = q->n[2];
  const uint64_t rr = q->nc[0], cc = q->nc[1], ff = q->nc[2];
  /* LPK1 along f: (nr,nc,nf) -> (nr,nc,ff); coarse^3 corner reads as zero
   * (zero_r,zero_c,zero_f = rr,cc,ff, LinearProcessingKernel3D.hpp:98-101). */
  REAL *t1 = (REAL *)malloc(sizeof(REAL) * nr * nc * ff);
  if (q->active[2]) {
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < nr; i++)
      for (uint64_t j = 0; j < nc; j++) {
        const REAL *row = v + IDX3(i, j, 0, ldv1, ldv2);
        FN(mass_trans_line)(nf, ff, q->dist[2], row, (i < rr && j < cc), row + ff, 1,
                            t1 + IDX3(i, j, 0, nc, ff), 1);
      }
  } else {
    free(t1);
    return NULL; /* f is always active (D>=1) */
  }
  /* LPK2 along c: (nr,nc,ff) -> (nr,cc,ff) */
  REAL *t2;
  if (q->active[1]) {
    t2 = (REAL *)malloc(sizeof(REAL) * nr * cc * ff);
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < nr; i++)
      for (uint64_t k = 0; k < ff; k++) {
        const REAL *col = t1 + IDX3(i, 0, k, nc, ff);
        FN(mass_trans_line)(nc, cc, q->dist[1], col, 0, col + cc * ff, ff,
                            t2 + IDX3(i, 0, k, cc, ff), ff);
      }
    free(t1);
  } else {
    t2 = t1; /* nc == cc == 1 */
  }
  /* LPK3 along r: (nr,cc,ff) -> (rr,cc,ff) */
  REAL *t3;
  if (q->active[0]) {
    t3 = (REAL *)malloc(sizeof(REAL) * rr * cc * ff);
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t j = 0; j < cc; j++)
      for (uint64_t k = 0; k < ff; k++) {
        const REAL *col = t2 + IDX3(0, j, k, cc, ff);
        FN(mass_trans_line)(nr, rr, q->dist[0], col, 0, col + rr * cc * ff, cc * ff,
                            t3 + IDX3(0, j, k, cc, ff), cc * ff);
      }
    free(t2);
  } else {
    t3 = t2;
  }
  /* IPK1 (f), IPK2 (c), IPK3 (r) in place */
#pragma omp parallel for collapse(2) schedule(static)
  for (uint64_t i = 0; i < rr; i++)
    for (uint64_t j = 0; j < cc; j++)
      FN(thomas_line)(ff, q->am[2], q->bm[2], t3 + IDX3(i, j, 0, cc, ff), 1);
  if (q->active[1]) {
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < rr; i++)
      for (uint64_t k = 0; k < ff; k++)
        FN(thomas_line)(cc, q->am[1], q->bm[1], t3 + IDX3(i, 0, k, cc, ff), ff);
  }
  if (q->active[0]) {
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t j = 0; j < cc; j++)
      for (uint64_t k = 0; k < ff; k++)
        FN(thomas_line)(rr, q->am[0], q->bm[0], t3 + IDX3(0, j, k, cc, ff), cc * ff);
  }
  return t3;
}

static void FN(full_strides)(const FN(mgxo_hier) * h, uint64_t *ld1, uint64_t *ld2,
                             uint64_t *n0) {
  int D = h->D;
  *ld2 = h->shape[D - 1];
  *ld1 = D >= 2 ? h->shape[D - 2] : 1;
  *n0 = D >= 3 ? h->shape[D - 3] : 1;
}

/* MultiDimension/DataRefactoring.hpp:25-109 (decompose, D<=3): in place on the
 * dense row-major array v of the finest shape. stop_level = 0. */
int FN(mgxo_decompose_nd)(const FN(mgxo_hier) * h, REAL *v);
int FN(mgxo_recompose_nd)(const FN(mgxo_hier) * h, REAL *v);
int FN(mgxo_decompose)(const FN(mgxo_hier) * h, REAL *v) {
  if (h->D > 3) return FN(mgxo_decompose_nd)(h, v);
  uint64_t ld1, ld2, n0;
  FN(full_strides)(h, &ld1, &ld2, &n0);
  for (int l = h->L; l > 0; l--) {
    FN(lvl3) q;
    FN(get_lvl3)(h, l, &q);
    const uint64_t nr = q.n[0], nc = q.n[1], nf = q.n[2];
    /* CopyND(v_fine -> w_fine) */
    REAL *w = (REAL *)malloc(sizeof(REAL) * nr * nc * nf);
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < nr; i++)
      for (uint64_t j = 0; j < nc; j++)
        memcpy(w + IDX3(i, j, 0, nc, nf), v + IDX3(i, j, 0, ld1, ld2), sizeof(REAL) * nf);
    FN(gpk_reo)(&q, w, v, ld1, ld2);
    free(w);
    REAL *corr = FN(calc_correction)(&q, v, ld1, ld2);
    /* AddND(w_correction -> v_coarse), LevelwiseProcessingKernel.hpp:69-74 */
    const uint64_t rr = q.nc[0], cc = q.nc[1], ff = q.nc[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < rr; i++)
      for (uint64_t j = 0; j < cc; j++)
        for (uint64_t k = 0; k < ff; k++)
          v[IDX3(i, j, k, ld1, ld2)] += corr[IDX3(i, j, k, cc, ff)];
    free(corr);
  }
  return 0;
}

/* MultiDimension/DataRefactoring.hpp:179-252 (recompose, D<=3). */
int FN(mgxo_recompose)(const FN(mgxo_hier) * h, REAL *v) {
  if (h->D > 3) return FN(mgxo_recompose_nd)(h, v);
  uint64_t ld1, ld2, n0;
  FN(full_strides)(h, &ld1, &ld2, &n0);
  for (int l = 1; l <= h->L; l++) {
    FN(lvl3) q;
    FN(get_lvl3)(h, l, &q);
    const uint64_t nr = q.n[0], nc = q.n[1], nf = q.n[2];
    const uint64_t rr = q.nc[0], cc = q.nc[1], ff = q.nc[2];
    REAL *corr = FN(calc_correction)(&q, v, ld1, ld2);
    /* SubtractND */
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < rr; i++)
      for (uint64_t j = 0; j < cc; j++)
        for (uint64_t k = 0; k < ff; k++)
          v[IDX3(i, j, k, ld1, ld2)] -= corr[IDX3(i, j, k, cc, ff)];
    free(corr);
    REAL *w = (REAL *)malloc(sizeof(REAL) * nr * nc * nf);
    FN(gpk_rev)(&q, v, ld1, ld2, w);
#pragma omp parallel for collapse(2) schedule(static)
    for (uint64_t i = 0; i < nr; i++)
      for (uint64_t j = 0; j < nc; j++)
        memcpy(v + IDX3(i, j, 0, ld1, ld2), w + IDX3(i, j, 0, nc, nf), sizeof(REAL) * nf);
    free(w);
  }
  return 0;
}

/* ======================================================================== */
/* N-D path (D = 4, 5; also valid for D <= 3, where it is cross-checked      */
/* against the 3-D code above).                                              */
/* Reference: MultiDimension/DataRefactoring.hpp:111-177 (decompose, D > 3), */
/* :254-317 (recompose); CalcCoefficientsND.hpp:25-236 -- interpolants are   */
/* built 3 dims at a time, fastest dims first (GpkReo<..,INTERPOLATION,..>   */
/* passes), i.e. nested lerps from the fastest to the slowest dim, then      */
/* coefficient = original - interpolant (the CALC_COEFF pass);               */
/* CalcCorrectionND.hpp:25-267 -- mass/restriction sweeps along dims         */
/* D-1, D-2, ..., 0 (the first one reads the all-coarse corner as zero:      */
/* zero_other && r<zero_r && c<zero_c && f<zero_f,                           */
/* LinearProcessingKernel.hpp:104-124) then Thomas solves in the same order. */
/* ======================================================================== */

static uint64_t FN(nd_prod)(int D, const uint64_t *e) {
  uint64_t p = 1;
  for (int d = 0; d < D; d++) p *= e[d];
  return p;
}
static void FN(nd_unravel)(int D, const uint64_t *e, uint64_t lin, uint64_t *idx) {
  for (int d = D - 1; d >= 0; d--) {
    idx[d] = lin % e[d];
    lin /= e[d];
  }
}
static uint64_t FN(nd_ravel)(int D, const uint64_t *e, const uint64_t *idx) {
  uint64_t lin = 0;
  for (int d = 0; d < D; d++) lin = lin * e[d] + idx[d];
  return lin;
}
/* fine position of reordered index i (coarse node -> min(2i, n-1); coefficient -> odd) */
static inline uint64_t FN(nd_fine_pos)(uint64_t i, uint64_t n, uint64_t m, int *odd) {
  *odd = i >= m;
  if (*odd) return 2 * (i - m) + 1;
  return 2 * i < n - 1 ? 2 * i : n - 1;
}

/* interpolant of the node at natural position pos[] (odd[] marks its odd dims) from its
 * even-index neighbours in the natural-order compact box w: nested lerps, fastest dim
 * innermost. Returns the node value itself if no dim is odd. */
static REAL FN(nd_interp)(const FN(mgxo_hier) * h, int l, const uint64_t *n, const REAL *w,
                          const uint64_t *pos, const int *odd) {
  const int D = h->D;
  int od[MGXO_MAXD], nod = 0;
  for (int d = D - 1; d >= 0; d--)
    if (odd[d]) od[nod++] = d; /* od[0] = fastest odd dim */
  REAL vals[1 << MGXO_MAXD];
  for (int c = 0; c < (1 << nod); c++) {
    uint64_t q[MGXO_MAXD];
    for (int d = 0; d < D; d++) q[d] = pos[d];
    for (int b = 0; b < nod; b++) q[od[b]] = pos[od[b]] + (((c >> b) & 1) ? 1 : -1);
    vals[c] = w[FN(nd_ravel)(D, n, q)];
  }
  for (int b = 0; b < nod; b++) {
    const REAL t = h->ratio[l][od[b]][pos[od[b]] - 1];
    const int cnt = 1 << (nod - b - 1);
    for (int c = 0; c < cnt; c++) vals[c] = FN(lerp)(vals[2 * c], vals[2 * c + 1], t);
  }
  return vals[0];
}

/* mass/restriction sweep along dim a: in has extents e (dim a: reordered [even | odd]), out has
 * the same extents with e[a] -> m. zero_all_coarse: the first sweep reads the even part of
 * lines whose other coordinates are all < mc[] as zero. */
static REAL *FN(nd_lpk)(int D, const uint64_t *e, int a, uint64_t n, uint64_t m, const REAL *dist,
                        const REAL *in, const uint64_t *in_strides, int zero_all_coarse,
                        const uint64_t *mc, uint64_t *eo) {
  for (int d = 0; d < D; d++) eo[d] = e[d];
  eo[a] = m;
  uint64_t el[MGXO_MAXD]; /* extents of the line index space (dim a collapsed) */
  for (int d = 0; d < D; d++) el[d] = e[d];
  el[a] = 1;
  const uint64_t nlines = FN(nd_prod)(D, el);
  REAL *out = (REAL *)malloc(sizeof(REAL) * FN(nd_prod)(D, eo));
  uint64_t os[MGXO_MAXD];
  {
    uint64_t sacc = 1;
    for (int d = D - 1; d >= 0; d--) {
      os[d] = sacc;
      sacc *= eo[d];
    }
  }
#pragma omp parallel for schedule(static)
  for (uint64_t ln = 0; ln < nlines; ln++) {
    uint64_t idx[MGXO_MAXD];
    FN(nd_unravel)(D, el, ln, idx);
    uint64_t ib = 0, ob = 0;
    int ez = zero_all_coarse;
    for (int d = 0; d < D; d++) {
      if (d == a) continue;
      ib += idx[d] * in_strides[d];
      ob += idx[d] * os[d];
      if (idx[d] >= mc[d]) ez = 0;
    }
    FN(mass_trans_line)(n, m, dist, in + ib, ez, in + ib + m * in_strides[a], in_strides[a],
                        out + ob, os[a]);
  }
  return out;
}

static void FN(nd_full_strides)(const FN(mgxo_hier) * h, uint64_t *fs) {
  uint64_t sacc = 1;
  for (int d = h->D - 1; d >= 0; d--) {
    fs[d] = sacc;
    sacc *= h->shape[d];
  }
}

/* correction of level l from the reordered coefficients in v (full strides fs) */
static REAL *FN(nd_correction)(const FN(mgxo_hier) * h, int l, const REAL *v, const uint64_t *fs) {
  const int D = h->D;
  uint64_t e[MGXO_MAXD], eo[MGXO_MAXD], mc[MGXO_MAXD];
  for (int d = 0; d < D; d++) {
    e[d] = h->lshape[l][d];
    mc[d] = h->lshape[l - 1][d];
  }
  const REAL *cur = v;
  uint64_t cs[MGXO_MAXD];
  for (int d = 0; d < D; d++) cs[d] = fs[d];
  REAL *owned = NULL;
  for (int a = D - 1; a >= 0; a--) {
    REAL *nx = FN(nd_lpk)(D, e, a, h->lshape[l][a], mc[a], h->dist[l][a], cur, cs,
                          a == D - 1, mc, eo);
    free(owned);
    owned = nx;
    cur = nx;
    for (int d = 0; d < D; d++) e[d] = eo[d];
    uint64_t sacc = 1;
    for (int d = D - 1; d >= 0; d--) {
      cs[d] = sacc;
      sacc *= e[d];
    }
  }
  /* Thomas solves along D-1 .. 0 on the compact coarse box */
  for (int a = D - 1; a >= 0; a--) {
    uint64_t el[MGXO_MAXD];
    for (int d = 0; d < D; d++) el[d] = e[d];
    el[a] = 1;
    const uint64_t nlines = FN(nd_prod)(D, el);
#pragma omp parallel for schedule(static)
    for (uint64_t ln = 0; ln < nlines; ln++) {
      uint64_t idx[MGXO_MAXD];
      FN(nd_unravel)(D, el, ln, idx);
      uint64_t b = 0;
      for (int d = 0; d < D; d++) b += idx[d] * cs[d];
      FN(thomas_line)(e[a], h->am[l - 1][a], h->bm[l - 1][a], owned + b, cs[a]);
    }
  }
  return owned;
}

int FN(mgxo_decompose_nd)(const FN(mgxo_hier) * h, REAL *v) {
  const int D = h->D;
  uint64_t fs[MGXO_MAXD];
  FN(nd_full_strides)(h, fs);
  for (int l = h->L; l > 0; l--) {
    uint64_t n[MGXO_MAXD], m[MGXO_MAXD];
    for (int d = 0; d < D; d++) {
      n[d] = h->lshape[l][d];
      m[d] = h->lshape[l - 1][d];
    }
    const uint64_t nn = FN(nd_prod)(D, n);
    REAL *w = (REAL *)malloc(sizeof(REAL) * nn);
#pragma omp parallel for schedule(static)
    for (uint64_t lin = 0; lin < nn; lin++) {
      uint64_t idx[MGXO_MAXD];
      FN(nd_unravel)(D, n, lin, idx);
      uint64_t off = 0;
      for (int d = 0; d < D; d++) off += idx[d] * fs[d];
      w[lin] = v[off];
    }
#pragma omp parallel for schedule(static)
    for (uint64_t lin = 0; lin < nn; lin++) {
      uint64_t idx[MGXO_MAXD], pos[MGXO_MAXD];
      int odd[MGXO_MAXD], any = 0;
      FN(nd_unravel)(D, n, lin, idx);
      uint64_t off = 0;
      for (int d = 0; d < D; d++) {
        pos[d] = FN(nd_fine_pos)(idx[d], n[d], m[d], &odd[d]);
        any |= odd[d];
        off += idx[d] * fs[d];
      }
      const REAL centre = w[FN(nd_ravel)(D, n, pos)];
      v[off] = any ? centre - FN(nd_interp)(h, l, n, w, pos, odd) : centre;
    }
    free(w);
    REAL *corr = FN(nd_correction)(h, l, v, fs);
    const uint64_t mm = FN(nd_prod)(D, m);
#pragma omp parallel for schedule(static)
    for (uint64_t lin = 0; lin < mm; lin++) {
      uint64_t idx[MGXO_MAXD];
      FN(nd_unravel)(D, m, lin, idx);
      uint64_t off = 0;
      for (int d = 0; d < D; d++) off += idx[d] * fs[d];
      v[off] += corr[lin];
    }
    free(corr);
  }
  return 0;
}

int FN(mgxo_recompose_nd)(const FN(mgxo_hier) * h, REAL *v) {
  const int D = h->D;
  uint64_t fs[MGXO_MAXD];
  FN(nd_full_strides)(h, fs);
  for (int l = 1; l <= h->L; l++) {
    uint64_t n[MGXO_MAXD], m[MGXO_MAXD];
    for (int d = 0; d < D; d++) {
      n[d] = h->lshape[l][d];
      m[d] = h->lshape[l - 1][d];
    }
    REAL *corr = FN(nd_correction)(h, l, v, fs);
    const uint64_t mm = FN(nd_prod)(D, m), nn = FN(nd_prod)(D, n);
#pragma omp parallel for schedule(static)
    for (uint64_t lin = 0; lin < mm; lin++) {
      uint64_t idx[MGXO_MAXD];
      FN(nd_unravel)(D, m, lin, idx);
      uint64_t off = 0;
      for (int d = 0; d < D; d++) off += idx[d] * fs[d];
      v[off] -= corr[lin];
    }
    free(corr);
    /* natural-order box: first the coarse nodes, then coefficient nodes */
    REAL *w = (REAL *)malloc(sizeof(REAL) * nn);
    for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for schedule(static)
      for (uint64_t lin = 0; lin < nn; lin++) {
        uint64_t idx[MGXO_MAXD], pos[MGXO_MAXD];
        int odd[MGXO_MAXD], any = 0;
        FN(nd_unravel)(D, n, lin, idx);
        uint64_t off = 0;
        for (int d = 0; d < D; d++) {
          pos[d] = FN(nd_fine_pos)(idx[d], n[d], m[d], &odd[d]);
          any |= odd[d];
          off += idx[d] * fs[d];
        }
        if (pass == 0 && !any) w[FN(nd_ravel)(D, n, pos)] = v[off];
        if (pass == 1 && any) {
          REAL res = v[off];
          res += FN(nd_interp)(h, l, n, w, pos, odd);
          w[FN(nd_ravel)(D, n, pos)] = res;
        }
      }
    }
#pragma omp parallel for schedule(static)
    for (uint64_t lin = 0; lin < nn; lin++) {
      uint64_t idx[MGXO_MAXD];
      FN(nd_unravel)(D, n, lin, idx);
      uint64_t off = 0;
      for (int d = 0; d < D; d++) off += idx[d] * fs[d];
      v[off] = w[lin];
    }
    free(w);
  }
  return 0;
}

/* Quantization/LinearQuantization.hpp:495-545 (CalcQuantizers). ebtype: 0=REL,
 * 1=ABS (Utilities/Types.h:32). MultiDim decomposition only. */
void FN(mgxo_calc_quantizers)(const FN(mgxo_hier) * h, int ebtype, REAL tol, REAL s, REAL norm,
                              int reciprocal, REAL *quantizers) {
  uint64_t dof = 1;
  for (int d = 0; d < h->D; d++) dof *= h->shape[d];
  double abs_tol = tol;
  if (ebtype == 0) abs_tol *= norm;
  abs_tol *= 2;
  uint64_t l_target = (uint64_t)h->L;
  if (s == (REAL)INFINITY) {
    for (int l = 0; l <= h->L; l++) {
      quantizers[l] = (REAL)((abs_tol) / ((l_target + 1) * (1 + pow(3, h->D))));
      if (reciprocal) quantizers[l] = 1.0f / quantizers[l];
    }
  } else {
    for (int l = 0; l <= h->L; l++) {
      quantizers[l] = (REAL)((abs_tol) / (EXP2R(s * l) * sqrt((double)dof)));
      if (reciprocal) quantizers[l] = 1.0f / quantizers[l];
    }
  }
}

/* Quantization/LinearQuantization.hpp:146-245 (quantize branch of
 * LevelwiseLinearQuantizerNDFunctor). Outliers are appended in index order here
 * (the reference's order is atomicAdd order, i.e. unspecified). Returns the
 * outlier count (may exceed outlier_cap; only the first outlier_cap are stored,
 * :233-236). */
uint64_t FN(mgxo_quantize)(const FN(mgxo_hier) * h, const REAL *v, int ebtype, REAL tol, REAL s,
                           REAL norm, uint64_t dict_size, int prep_huffman, int64_t *qv,
                           uint64_t *outlier_idx, int64_t *outlier_val, uint64_t outlier_cap) {
  REAL quantizers[MGXO_MAXL + 1];
  FN(mgxo_calc_quantizers)(h, ebtype, tol, s, norm, 1, quantizers);
  const int calc_vol = !(s == (REAL)INFINITY);
  const int D = h->D;
  uint64_t total = 1;
  for (int d = 0; d < D; d++) total *= h->shape[d];
  int nt_max = 1;
#ifdef _OPENMP
  nt_max = omp_get_max_threads();
#endif
  uint64_t **lidx = (uint64_t **)calloc(nt_max, sizeof(*lidx));
  int64_t **lval = (int64_t **)calloc(nt_max, sizeof(*lval));
  uint64_t *lcnt = (uint64_t *)calloc(nt_max, sizeof(*lcnt));
#pragma omp parallel
  {
    int t = 0, nt = 1;
#ifdef _OPENMP
    t = omp_get_thread_num();
    nt = omp_get_num_threads();
#endif
    uint64_t lo = total / nt * t + (total % nt < (uint64_t)t ? total % nt : (uint64_t)t);
    uint64_t hi = lo + total / nt + ((uint64_t)t < total % nt ? 1 : 0);
    uint64_t cap = 0, cnt = 0;
    uint64_t *li = NULL;
    int64_t *lv = NULL;
    for (uint64_t lin = lo; lin < hi; lin++) {
      uint64_t idx[MGXO_MAXD], rem = lin;
      for (int d = D - 1; d >= 0; d--) {
        idx[d] = rem % h->shape[d];
        rem /= h->shape[d];
      }
      REAL tv = v[lin];
      REAL volume = 1;
      int level = 0;
      if (calc_vol) {
        for (int d = D - 1; d >= 0; d--)
          if (h->marks[d][idx[d]] > level) level = h->marks[d][idx[d]];
        for (int d = D - 1; d >= 0; d--) volume *= h->vol[level][d];
        volume = SQRTR(volume);
      }
      REAL quantizer = quantizers[level];
      int64_t qd = (int64_t)COPYSIGNR((REAL)0.5 + FABSR(tv * quantizer * volume), tv);
      if (prep_huffman) {
        qd += (int64_t)(dict_size / 2);
        if (!(qd >= 0 && qd < (int64_t)dict_size)) {
          if (cnt == cap) {
            cap = cap ? cap * 2 : 1024;
            li = (uint64_t *)realloc(li, cap * sizeof(*li));
            lv = (int64_t *)realloc(lv, cap * sizeof(*lv));
          }
          li[cnt] = lin;
          lv[cnt] = qd;
          cnt++;
          qd = 0;
        }
      }
      qv[lin] = qd;
    }
    lidx[t] = li;
    lval[t] = lv;
    lcnt[t] = cnt;
  }
  uint64_t count = 0;
  for (int t = 0; t < nt_max; t++) {
    for (uint64_t i = 0; i < lcnt[t]; i++) {
      if (count < outlier_cap) {
        outlier_idx[count] = lidx[t][i];
        outlier_val[count] = lval[t][i];
      }
      count++;
    }
    free(lidx[t]);
    free(lval[t]);
  }
  free(lidx);
  free(lval);
  free(lcnt);
  return count;
}

/* Quantization/LinearQuantization.hpp:304-350 (OutlierRestore) + :246-264
 * (dequantize branch). qv is modified (outliers restored) like the reference. */
void FN(mgxo_dequantize)(const FN(mgxo_hier) * h, int64_t *qv, int ebtype, REAL tol, REAL s,
                         REAL norm, uint64_t dict_size, int prep_huffman,
                         const uint64_t *outlier_idx, const int64_t *outlier_val,
                         uint64_t outlier_count, REAL *v) {
  REAL quantizers[MGXO_MAXL + 1];
  FN(mgxo_calc_quantizers)(h, ebtype, tol, s, norm, 0, quantizers);
  const int calc_vol = !(s == (REAL)INFINITY);
  const int D = h->D;
  uint64_t total = 1;
  for (int d = 0; d < D; d++) total *= h->shape[d];
  if (prep_huffman)
    for (uint64_t i = 0; i < outlier_count; i++) qv[outlier_idx[i]] = outlier_val[i];
#pragma omp parallel for schedule(static)
  for (uint64_t lin = 0; lin < total; lin++) {
    uint64_t idx[MGXO_MAXD], rem = lin;
    for (int d = D - 1; d >= 0; d--) {
      idx[d] = rem % h->shape[d];
      rem /= h->shape[d];
    }
    REAL volume = 1;
    int level = 0;
    if (calc_vol) {
      for (int d = D - 1; d >= 0; d--)
        if (h->marks[d][idx[d]] > level) level = h->marks[d][idx[d]];
      for (int d = D - 1; d >= 0; d--) volume *= h->vol_recip[level][d];
      volume = SQRTR(volume);
    }
    REAL quantizer = quantizers[level];
    int64_t qd = qv[lin];
    if (prep_huffman) qd -= (int64_t)(dict_size / 2);
    v[lin] = (quantizer * volume) * (REAL)qd;
  }
}

/* config.reorder == 1 ("level linearised" quantized output): position of the element with
 * reordered N-D linear index `lin` in the 1-D array whose slot for level l starts where the
 * levels below it end (Quantization/LinearQuantization.hpp:588-605: level_size(l) - level_size(l-1)
 * elements behind quantized_data(last_level_size)), at the offset calc_level_offset() gives it
 * inside that slot (:46-146, restated statement by statement; the region offsets it also
 * computes are not used by its result). */
uint64_t FN(mgxo_linearized_position)(const FN(mgxo_hier) * h, uint64_t lin) {
  const int D = h->D;
  uint64_t idx[MGXO_MAXD], rem = lin;
  for (int d = D - 1; d >= 0; d--) {
    idx[d] = rem % h->shape[d];
    rem /= h->shape[d];
  }
  int level = 0;
  for (int d = D - 1; d >= 0; d--)
    if (h->marks[d][idx[d]] > level) level = h->marks[d][idx[d]]; /* :47-49 */
  uint64_t curr_region = 0;
  for (int d = D - 1; d >= 0; d--)
    curr_region += (uint64_t)(level == h->marks[d][idx[d]]) << d; /* :51-55 */
  /* level_ranges(l, d) = shape of level l-1 (0 for l = 0), Hierarchy.hpp:240-259 */
  uint64_t coarse_level_size[MGXO_MAXD], fine[MGXO_MAXD];
  for (int d = D - 1; d >= 0; d--) {
    coarse_level_size[d] = level == 0 ? 0 : h->lshape[level - 1][d]; /* level_ranges(level, d) */
    fine[d] = h->lshape[level][d];                                   /* level_ranges(level + 1, d) */
  }
  uint64_t curr_region_thread_idx[MGXO_MAXD], global_data_idx[MGXO_MAXD];
  for (int d = D - 1; d >= 0; d--) {
    uint64_t bit = (curr_region >> d) & 1u;
    curr_region_thread_idx[d] = bit ? idx[d] - coarse_level_size[d] : idx[d]; /* :92-96 */
  }
  for (int d = D - 1; d >= 0; d--) { /* :98-110 */
    uint64_t bit = (curr_region >> d) & 1u;
    if (level == 0) {
      global_data_idx[d] = curr_region_thread_idx[d];
    } else if (fine[d] % 2 == 0 && curr_region_thread_idx[d] == fine[d] / 2) {
      global_data_idx[d] = fine[d] - 1;
    } else {
      global_data_idx[d] = curr_region_thread_idx[d] * 2 + bit;
    }
  }
  uint64_t stride = 1, curr_thread_offset = 0, coarse_level_offset = 0;
  for (int d = D - 1; d >= 0; d--) { /* :112-116 */
    curr_thread_offset += global_data_idx[d] * stride;
    stride *= fine[d];
  }
  stride = 1;
  for (int d = D - 1; d >= 0; d--) { /* :118-128 */
    if (global_data_idx[d] % 2 != 0 && global_data_idx[d] != fine[d] - 1) coarse_level_offset = 0;
    if (global_data_idx[d]) coarse_level_offset += ((global_data_idx[d] - 1) / 2 + 1) * stride;
    stride *= fine[d] / 2 + 1;
  }
  if (level == 0) coarse_level_offset = 0;
  uint64_t base = 0;
  if (level > 0) {
    base = 1;
    for (int d = 0; d < D; d++) base *= h->lshape[level - 1][d];
  }
  return base + (curr_thread_offset - coarse_level_offset);
}

/* out[position(lin)] = in[lin] (inverse: out[lin] = in[position(lin)]) */
void FN(mgxo_level_linearize)(const FN(mgxo_hier) * h, const int64_t *in, int64_t *out, int inverse) {
  uint64_t total = 1;
  for (int d = 0; d < h->D; d++) total *= h->shape[d];
#pragma omp parallel for schedule(static)
  for (uint64_t lin = 0; lin < total; lin++) {
    const uint64_t p = FN(mgxo_linearized_position)(h, lin);
    if (inverse) out[lin] = in[p];
    else out[p] = in[lin];
  }
}

/* CompressionLowLevel/NormCalculator.hpp:12-80. s=inf: max |x|; else
 * sqrt(sum x^2 [/N]) with the SERIAL backend's sequential accumulation in T
 * (RuntimeX/DeviceAdapters/DeviceAdapterSerial.h:1376-1381). 0 -> epsilon. */
REAL FN(mgxo_norm)(const REAL *v, uint64_t n, REAL s, int normalize_coordinates) {
  REAL norm = 0;
  if (s == (REAL)INFINITY) {
    for (uint64_t i = 0; i < n; i++) {
      REAL a = FABSR(v[i]);
      if (a > norm) norm = a;
    }
  } else {
    REAL acc = 0;
    for (uint64_t i = 0; i < n; i++) acc += v[i] * v[i];
    norm = normalize_coordinates ? SQRTR(acc / (REAL)n) : SQRTR(acc);
  }
  if (norm == 0) norm = EPSR;
  return norm;
}

#undef IDX3
#undef FN
#undef CAT
#undef CAT_
