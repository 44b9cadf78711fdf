"""Pin the oracle's OPERATORS -- interpolation, mass matrix x restriction, mass solve -- on
NON-UNIFORM spacing with the known answers the reference's own operator tests hold
(tests/golden/reference_goldens.json["operators"], extracted by
tests/golden/extract_reference_goldens.py from the reference's
tests/src/test_TensorMassMatrix.cpp:21-262, test_TensorRestriction.cpp:18-221,
test_TensorProlongation.cpp:16-106; dyadic sections only -- on those MGARD-CPU's hierarchy is
MGARD-X's).

The chain of evidence:
  1. the three operators written down here as dense matrices of the node coordinates (P1 mass
     matrix, linear interpolation P, restriction P^T) reproduce EVERY reference vector, default
     and custom spacing -- so these matrices are the reference's operators;
  2. the oracle's line operators (the very functions its decomposition calls: lerp
     GPKFunctor.h:21-23, mass_trans LPKFunctor.h:77-93, the Thomas solve IPKFunctor.h:111-149 with
     the am / bm of Hierarchy.hpp:112-162) equal those matrices on the reference's coordinate sets
     and inputs -- and R applied to the reference's own M u equals the oracle's fused R M u;
  3. a whole decomposition assembled from the matrices -- per level: coefficients u - P u_coarse,
     correction (x) M_c^-1 P^T M_f over the dimensions -- equals oracle.decompose on non-uniform
     dyadic grids in 1-3 D (the reference's 5 x 5 custom spacing among them), float and double.
The -m gpu mirror of step 3 through mgh_decompose is tests/test_gpu_parity.py::
test_decompose_equals_the_operator_built_decomposition_on_nonuniform_grids."""
import json
import os

import numpy as np
import pytest

import oracle

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))
OPS = G["operators"]["cases"]
DT = {"float": np.float32, "double": np.float64}


# ---- the operators as dense matrices (float64) of the node coordinates x ----------------------
def mass_matrix(x):
    """P1 mass matrix of the mesh x: (h_{i-1} + h_i) / 3 on the diagonal, h_i / 6 beside it."""
    x = np.asarray(x, dtype=np.float64)
    n, h = x.size, np.diff(x)
    M = np.zeros((n, n))
    for i in range(n - 1):
        M[i, i] += h[i] / 3
        M[i + 1, i + 1] += h[i] / 3
        M[i, i + 1] += h[i] / 6
        M[i + 1, i] += h[i] / 6
    return M


def prolongation(x):
    """Linear interpolation from the even nodes of x to all of them."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    assert n % 2 == 1
    P = np.zeros((n, n // 2 + 1))
    for p in range(n):
        if p % 2 == 0:
            P[p, p // 2] = 1
        else:
            t = (x[p] - x[p - 1]) / (x[p + 1] - x[p - 1])
            P[p, p // 2], P[p, p // 2 + 1] = 1 - t, t
    return P


def coords_of(case):
    if case["coords"] is not None:
        return [np.asarray(c, dtype=np.float64) for c in case["coords"]]
    return [np.arange(n, dtype=np.float64) / (n - 1) for n in case["shape"]]  # TensorMeshHierarchy default


def level_nodes(n, L, l):
    return np.arange(0, n, 1 << (L - l))


def levels_of(shape):
    return min((n - 1).bit_length() - 1 for n in shape)


def apply_line(op, x_line, v_line):
    """One constituent operator on the level nodes of a line, as the reference's Constituent*
    classes apply it in place (TensorMassMatrix.tpp:15-90, TensorRestriction.tpp:23-71,
    TensorProlongation.tpp:22-69)."""
    v = np.array(v_line, dtype=np.float64)
    if op == "mass":
        return mass_matrix(x_line) @ v
    P = prolongation(x_line)
    if op == "restriction":      # coarse nodes += P^T (values on the new nodes)
        new = np.zeros_like(v)
        new[1::2] = v[1::2]
        v[0::2] += (P.T @ new)
        return v
    if op == "prolongation_addition":  # new nodes += interpolant of the coarse nodes
        v[1::2] += (P @ v[0::2])[1::2]
        return v
    raise ValueError(op)


def apply_case(case):
    shape = tuple(case["shape"])
    x = coords_of(case)
    L = levels_of(shape)
    u = np.array(case["u"], dtype=np.float64).reshape(shape)
    l = case["l"]
    if case["op"].startswith("tensor_"):
        op = case["op"][len("tensor_"):]
        sel = tuple(level_nodes(n, L, l) for n in shape)
        sub = u[np.ix_(*sel)].copy()
        for d in range(len(shape)):
            sub = np.apply_along_axis(lambda line: apply_line(op, x[d][sel[d]], line), d, sub)
        u[np.ix_(*sel)] = sub
        return u
    d = case["dimension"]
    nodes = level_nodes(shape[d], L, l)
    for mi in case["multiindices"]:
        idx = list(mi)
        idx[d] = nodes
        idx = tuple(idx)
        u[idx] = apply_line(case["op"], x[d][nodes], u[idx])
    return u


# ---- 1. the matrices reproduce every reference vector ----------------------------------------
@pytest.mark.parametrize("k", range(len(OPS)))
def test_matrices_reproduce_the_reference_vectors(k):
    case = OPS[k]
    got = apply_case(case).ravel()
    want = np.array(case["expected"], dtype=np.float64)
    # (the reference compares with Catch::Approx, epsilon 100 * FLT_EPSILON; the literals are
    # exact rationals, so double arithmetic reproduces them to rounding)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (case["source"], case["op"], case["l"])


def test_the_vectors_cover_custom_spacing_and_every_operator():
    kinds = {(c["op"], c["coords"] is not None) for c in OPS}
    for op in ("mass", "restriction", "prolongation_addition"):
        assert (op, True) in kinds and (op, False) in kinds
    assert ("tensor_mass", True) in kinds and ("tensor_restriction", True) in kinds


# ---- 2. the oracle's line operators equal the matrices ---------------------------------------
def _coordinate_sets():
    seen, out = set(), []
    for c in OPS:
        for x in coords_of(c):
            key = tuple(x)
            if len(x) >= 3 and key not in seen:
                seen.add(key)
                out.append(np.asarray(x))
    rng = np.random.default_rng(3)
    for n in (9, 17, 33):   # (more levels than the reference's 3- and 5-node sets give)
        out.append(np.cumsum(rng.uniform(0.05, 1.0, size=n)))
    return out


def _inputs(n, k):
    rng = np.random.default_rng(100 + k)
    yield rng.integers(-10, 11, size=n).astype(np.float64)   # (like the reference's literals)
    yield rng.normal(size=n)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_oracle_line_operators_equal_the_matrices(dt):
    tol = 2e-6 if dt == np.float32 else 1e-13
    checked = 0
    for k, x in enumerate(_coordinate_sets()):
        n = x.size
        h = oracle.Hierarchy((n,), dt, coords=[x.astype(dt)])
        xs = x.astype(dt).astype(np.float64)  # the coordinates the oracle actually sees
        L = h.l_target
        assert L == levels_of((n,))
        for l in range(L, 0, -1):
            nodes = level_nodes(n, L, l)
            xl = xs[nodes]
            M, P = mass_matrix(xl), prolongation(xl)
            Mc = mass_matrix(xl[::2])
            for v in _inputs(nodes.size, k):
                vd = v.astype(dt).astype(np.float64)
                scale = np.abs(vd).max() + 1
                lerp = h.op_lerp(l, 0, vd.astype(dt)).astype(np.float64)
                assert np.allclose(lerp, (P @ vd[::2])[1::2], rtol=tol, atol=tol * scale)
                mt = h.op_mass_trans(l, 0, vd.astype(dt)).astype(np.float64)
                want = P.T @ (M @ vd)
                assert np.allclose(mt, want, rtol=tol, atol=tol * np.abs(want).max())
                rhs = Mc @ vd[::2]
                sol = h.op_thomas(l - 1, 0, rhs.astype(dt)).astype(np.float64)
                # (the solve amplifies the rounding of the right-hand side by the condition of M_c)
                cond = np.linalg.cond(Mc)
                assert np.allclose(sol, vd[::2], rtol=tol * cond, atol=tol * cond * scale)
                checked += 1
    assert checked >= 20


def test_oracle_mass_trans_on_the_references_own_mass_vectors():
    """R applied to the vector the reference expects for M u equals the oracle's fused R M u: the
    restriction of `expected` is taken with the matrix that step 1 pinned, M u is the reference's
    number itself (custom spacing, test_TensorMassMatrix.cpp:96-213)."""
    done = 0
    for case in OPS:
        if case["op"] != "mass" or case["l"] < 1:
            continue
        shape = tuple(case["shape"])
        x = coords_of(case)
        L = levels_of(shape)
        d, l = case["dimension"], case["l"]
        nodes = level_nodes(shape[d], L, l)
        u = np.array(case["u"], dtype=np.float64).reshape(shape)
        e = np.array(case["expected"], dtype=np.float64).reshape(shape)
        h = oracle.Hierarchy((shape[d],), np.float64, coords=[x[d]])
        P = prolongation(x[d][nodes])
        for mi in case["multiindices"]:
            idx = list(mi)
            idx[d] = nodes
            idx = tuple(idx)
            got = h.op_mass_trans(l, 0, u[idx])
            assert np.allclose(got, P.T @ e[idx], rtol=1e-12, atol=1e-14), case["source"]
            done += 1
    assert done >= 6


# ---- 3. a decomposition assembled from the matrices == oracle.decompose ----------------------
def decompose_by_matrices(coords, u):
    """Multilevel coefficients in natural node order: per level l = L..1 on the level-l grid,
    coefficients = u - (x)P u_coarse on the new nodes, coarse nodes += (x)(M_c^-1 P^T M_f) of
    the coefficient field (DataRefactoring.hpp:80-109 in exact arithmetic)."""
    u = np.array(u, dtype=np.float64)
    shape = u.shape
    L = levels_of(shape)
    for l in range(L, 0, -1):
        sel = tuple(level_nodes(n, L, l) for n in shape)
        fine = u[np.ix_(*sel)].copy()
        coarse = fine[tuple(slice(None, None, 2) for _ in shape)]
        interp = coarse
        for d in range(len(shape)):
            P = prolongation(np.asarray(coords[d], dtype=np.float64)[sel[d]])
            interp = np.moveaxis(np.tensordot(P, interp, axes=(1, d)), 0, d)
        coef = fine - interp                      # zero on the coarse nodes
        corr = coef
        for d in range(len(shape)):
            xl = np.asarray(coords[d], dtype=np.float64)[sel[d]]
            A = np.linalg.solve(mass_matrix(xl[::2]), prolongation(xl).T @ mass_matrix(xl))
            corr = np.moveaxis(np.tensordot(A, corr, axes=(1, d)), 0, d)
        fine = coef
        fine[tuple(slice(None, None, 2) for _ in shape)] = coarse + corr
        u[np.ix_(*sel)] = fine
    return u


NONUNIFORM_GRIDS = [
    ((5, 5), "reference"),          # the custom spacing of test_TensorMassMatrix.cpp:97-98
    ((9,), 1), ((33,), 2), ((17, 9), 3), ((9, 17), 4), ((9, 5, 9), 5), ((5, 9, 17), 6), ((17, 17, 17), 7),
]


def nonuniform_grid(shape, seed):
    if seed == "reference":
        c = next(c for c in OPS if c["shape"] == [5, 5] and c["coords"] is not None)
        return [np.asarray(x, dtype=np.float64) for x in c["coords"]]
    rng = np.random.default_rng(seed)
    return [np.cumsum(rng.uniform(0.1, 1.0, size=n)) for n in shape]


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape,seed", NONUNIFORM_GRIDS)
def test_oracle_decompose_equals_the_operator_built_decomposition(shape, seed, dt):
    coords = [x.astype(dt) for x in nonuniform_grid(shape, seed)]
    rng = np.random.default_rng(11)
    u = rng.normal(size=shape).astype(dt)
    h = oracle.Hierarchy(shape, dt, coords=coords)
    got = oracle.dyadic_reordered_to_natural(h.decompose(u)).astype(np.float64)
    want = decompose_by_matrices([c.astype(np.float64) for c in coords], u.astype(np.float64))
    tol = 5e-5 if dt == np.float32 else 1e-11
    assert np.allclose(got, want, rtol=tol, atol=tol * np.abs(want).max()), np.abs(got - want).max()
    # and back
    back = h.recompose(h.decompose(u))
    assert np.allclose(back, u, rtol=tol, atol=tol)


# ---- the quantizer's reference-held PROPERTIES (no vectors exist for it) ---------------------
@pytest.mark.parametrize("shape", [(129,), (23, 17), (8, 8, 9)])
@pytest.mark.parametrize("s", [np.inf, -0.75, 0.0, 1.5])
@pytest.mark.parametrize("tol", [0.001, 0.1, 1.0])
def test_quantization_inverts_dequantization(shape, s, tol):
    """tests/src/test_TensorMultilevelCoefficientQuantizer.cpp:59-207 ("(de)quantization
    inversion": shapes {129}, {23, 17}, {8, 8, 9}, s in {inf, -0.75, 0, 1.5}, tolerances {0.001,
    0.1, 1}, integers in [-1000, 1000]): quantize(dequantize(n)) == n, here for the MGARD-X
    level-wise quantizer (LinearQuantization.hpp:146-264)."""
    rng = np.random.default_rng(5)
    h = oracle.Hierarchy(shape, np.float32)
    n = rng.integers(-1000, 1001, size=shape).astype(np.int64)
    x = h.dequantize(n, oracle.ABS, np.float32(tol), np.float32(s), np.float32(1), prep_huffman=False)
    q, _, _, _ = h.quantize(x, oracle.ABS, np.float32(tol), np.float32(s), np.float32(1), prep_huffman=False)
    assert np.array_equal(q, n)
