/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for the MGARD-X (mgard_x::) multilevel decomposition + level-wise
 * linear quantizer hot path: a plain-C restatement of the reference algorithm
 * (reference: CODARcode/MGARD v1.6.0, include/mgard-x/...; each function in
 * mgx_oracle_impl.h cites the file:line it follows).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product (mgard_amd/, include/) never links, imports or calls it.
 *
 * PINNING STATUS: pinned against the reference's own golden vectors
 * (tests/src/test_decompose.cpp:276-457 decomposition, :549-838 recomposition;
 * committed as tests/golden/reference_goldens.json by
 * tests/golden/extract_reference_goldens.py) on dyadic (2^k+1) grids, where
 * MGARD-CPU and MGARD-X agree up to the level permutation and float rounding.
 * Also pinned (tests/test_oracle_goldens.py): one EVEN-size case (the ghost-node
 * rule) derived from the 1-D golden through the even = odd embedding, and the
 * scalar quantizer's known answers (tests/src/test_LinearQuantizer.cpp:94-111).
 * NON-UNIFORM spacing and the operators themselves (tests/test_oracle_operator_goldens.py):
 * the known answers of the reference's mass-matrix / restriction / prolongation tests on
 * default and custom spacing (tests/src/test_TensorMassMatrix.cpp:21-262,
 * test_TensorRestriction.cpp:18-221, test_TensorProlongation.cpp:16-106) pin dense-matrix forms
 * of the three operators; the line operators below (test hooks mgxo_op_*) equal those matrices,
 * and a decomposition assembled from them equals mgxo_decompose on non-uniform dyadic grids.
 * The MGARD-X SERIAL backend itself cannot be built under this repo's rules
 * (it needs the cmake-generated MGARDXConfig.h and zstd headers that are not on
 * the system include path), so for general NON-dyadic 2-D / 3-D shapes (the
 * ghost-node rule beyond that one case) and for the level-dependent quantizers
 * (s != infinity) parity is "unpinned": it rests
 * on the code reading cited in mgx_oracle_impl.h plus structural property tests
 * (tests/test_oracle_properties.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MGXO_MAXD 5
#define MGXO_MAXL 40

#define REAL float
#define SUFFIX _f32
#define SQRTR sqrtf
#define FABSR fabsf
#define COPYSIGNR copysignf
#define EXP2R exp2f
#define EPSR FLT_EPSILON
#include "mgx_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SQRTR
#undef FABSR
#undef COPYSIGNR
#undef EXP2R
#undef EPSR

#define REAL double
#define SUFFIX _f64
#define SQRTR sqrt
#define FABSR fabs
#define COPYSIGNR copysign
#define EXP2R exp2
#define EPSR DBL_EPSILON
#include "mgx_oracle_impl.h"

int mgxo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void mgxo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
