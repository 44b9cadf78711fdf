"""Pin the CPU oracle (oracle/) against the reference's own golden vectors
(tests/golden/reference_goldens.json, extracted from the reference's
tests/src/test_decompose.cpp by tests/golden/extract_reference_goldens.py).

The reference goldens are MGARD-CPU results in natural node order; on dyadic (2^k+1) uniform
grids MGARD-X computes the same multilevel coefficients up to its in-place level reordering and
float rounding, so they pin the MGARD-X restatement too. Tolerance is the
reference test's own: Catch::Approx(expected).epsilon(1e-4) (test_decompose.cpp:59)."""
import json
import os

import numpy as np
import pytest

import oracle

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))
CASES = {(c["kind"], c["name"]): c for c in G["cases"]}
DT = {"float": np.float32, "double": np.float64}


def approx(actual, expected, eps=1e-4):
    # Catch::Approx: |a-e| <= eps * (scale + |e|) with scale = 0 ... plus its default margin 0;
    # Catch2 v3 uses margin 0 and epsilon*|e|; exact zeros in the goldens need a small margin.
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    return np.all(np.abs(actual - expected) <= eps * np.abs(expected) + 1e-5)


natural_to_reordered = oracle.dyadic_natural_to_reordered
reordered_to_natural = oracle.dyadic_reordered_to_natural


@pytest.mark.parametrize("name", ["1D, dyadic, uniform", "2D, dyadic, uniform"])
def test_decomposition_goldens(name):
    c = CASES[("decomposition", name)]
    nd, dt = c["ndim"], DT[c["dtype"]]
    checked = 0
    for L, expected in enumerate(c["expecteds"]):
        n = (1 << L) + 1
        if n < 3:
            continue  # mgard_x::Hierarchy rejects dims < 3 (Hierarchy.hpp:742-756)
        shape = (n,) * nd
        u = np.array(c["u"][: n ** nd], dtype=dt).reshape(shape)
        h = oracle.Hierarchy(shape, dt, normalize_coordinates=False)
        assert h.l_target == L
        got = reordered_to_natural(h.decompose(u))
        assert approx(got.ravel(), expected), (name, L)
        # normalised coordinates change dist by a constant factor only: same coefficients
        h2 = oracle.Hierarchy(shape, dt, normalize_coordinates=True)
        assert approx(reordered_to_natural(h2.decompose(u)).ravel(), expected), (name, L)
        checked += 1
    assert checked >= 2


def test_decomposition_nonuniform_golden():
    c = CASES[("decomposition", "1D, dyadic, nonuniform")]
    coords = [np.array(c["coordinates"][0][0], dtype=np.float32)]
    h = oracle.Hierarchy((3,), np.float32, coords=coords)
    got = reordered_to_natural(h.decompose(np.array(c["u"], dtype=np.float32)))
    assert approx(got, c["expected"])


@pytest.mark.parametrize("name", ["1D, dyadic, uniform", "3D, dyadic, uniform"])
def test_recomposition_goldens(name):
    c = CASES[("recomposition", name)]
    nd, dt = c["ndim"], DT[c["dtype"]]
    checked = 0
    for L, expected in enumerate(c["expecteds"]):
        n = (1 << L) + 1
        if n < 3:
            continue
        shape = (n,) * nd
        u = np.array(c["u"][: n ** nd], dtype=dt).reshape(shape)
        h = oracle.Hierarchy(shape, dt, normalize_coordinates=False)
        got = h.recompose(natural_to_reordered(u))
        assert approx(got.ravel(), expected), (name, L)
        checked += 1
    assert checked >= 2


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(33,), (17, 9), (9, 5, 17), (6,), (8, 12), (10, 7, 12), (20, 17, 33)])
def test_decompose_recompose_roundtrip(shape, dt):
    rng = np.random.default_rng(1234)
    u = rng.standard_normal(shape).astype(dt)
    h = oracle.Hierarchy(shape, dt)
    back = h.recompose(h.decompose(u))
    tol = 2e-5 if dt == np.float32 else 1e-13
    assert np.max(np.abs(back - u)) <= tol * max(1.0, np.max(np.abs(u))) * 10


@pytest.mark.parametrize("kind", ["decomposition", "recomposition"])
def test_4d_goldens(kind):
    """The reference's 4-D goldens (3^4, test_decompose.cpp:339-431, :652-746) through the N-D
    restatement (the D > 3 path of MGARD-X: CalcCoefficientsND / CalcCorrectionND)."""
    c = CASES[(kind, "4D, dyadic, uniform")]
    u = np.array(c["u"][:81], dtype=np.float32).reshape((3,) * 4)
    h = oracle.Hierarchy((3,) * 4, np.float32, normalize_coordinates=False)
    assert h.l_target == 1
    if kind == "decomposition":
        got = reordered_to_natural(h.decompose(u))
    else:
        got = h.recompose(natural_to_reordered(u))
    assert approx(got.ravel(), c["expecteds"][1])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(9,), (6, 9), (9, 6, 8), (17, 20, 33), (6, 6, 6)])
def test_nd_restatement_equals_3d_restatement(shape, dt):
    rng = np.random.default_rng(3)
    u = rng.standard_normal(shape).astype(dt)
    h = oracle.Hierarchy(shape, dt)
    a, b = h.decompose(u), h.decompose(u, force_nd=True)
    assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))
    assert np.array_equal(h.recompose(a), h.recompose(a, force_nd=True))


@pytest.mark.parametrize("shape", [(5, 5, 5, 5), (6, 9, 8, 12), (3, 4, 5, 6, 7)])
def test_nd_roundtrip(shape):
    rng = np.random.default_rng(5)
    u = rng.standard_normal(shape)
    h = oracle.Hierarchy(shape, np.float64)
    assert np.max(np.abs(h.recompose(h.decompose(u)) - u)) < 1e-12


# ---- an EVEN size tied to a reference-held number -------------------------------------------------
# MGARD-X treats an even dim n as the odd dim n + 1 whose extra ("ghost") node sits mid-way in the
# last cell and has a zero coefficient (Hierarchy.hpp:38-42). Take the 4-node grid x = (0, 1, 2, 4):
# its ghost node is x = 3, so the embedded odd grid is the UNIFORM 5-node grid of the reference's
# golden "1D, dyadic, uniform" (test_decompose.cpp:277-338), input u* = (10, 3, -8, -6, 3), expected
# D(u*) = (4.4375, 2, -14.5, -3.5, -6.6875). The even grid carries (u0, u1, u2, u4) = (10, 3, -8, 3);
# its ghost value is the interpolant (u2 + u4) / 2 = -2.5 = u*_3 + 3.5, and the decomposition is
# linear, so   D_even = D(u*) + 3.5 D(e3)   on the four real nodes, where D(e3) for the unit vector
# at node 3 follows by hand: level-2 coefficients (0, 1) at nodes 1, 3; load vector of the coarse
# nodes (0, 1/2, 1/2); coarse mass matrix (2/6) [[2,1,0],[1,4,1],[0,1,2]] => correction
# (-1/8, 1/4, 5/8); the level-1 coefficient of node 2 is 1/4 - (-1/8 + 5/8) / 2 = 0:
#   D(e3) = (-1/8, 0, 0, 1, 5/8).
# Hence D_even + ghost = (4.0, 2.0, -14.5, 0.0, -4.5): the ghost coefficient comes out as exactly 0
# -- the rule itself -- and the four real nodes are reference-held numbers plus that closed form.
def _even4_expected():
    c = CASES[("decomposition", "1D, dyadic, uniform")]
    ustar = np.array(c["u"][:5], dtype=np.float64)
    d_ustar = np.array(c["expecteds"][2], dtype=np.float64)
    d_e3 = np.array([-0.125, 0.0, 0.0, 1.0, 0.625])
    delta = (ustar[2] + ustar[4]) / 2 - ustar[3]
    full = d_ustar + delta * d_e3
    assert abs(full[3]) < 1e-12          # the ghost node's coefficient vanishes
    x_even = ustar[[0, 1, 2, 4]]
    # MGARD-X layout of n = 4 (levels 4 -> 3 -> 2): [node 0, node 3 | level-1 coefficient
    # (node 2) | level-2 coefficient (node 1)]
    reordered = np.array([full[0], full[4], full[2], full[1]])
    return x_even, reordered


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_even_size_ghost_node_against_reference_number(dt):
    x_even, expected = _even4_expected()
    h = oracle.Hierarchy((4,), dt, coords=[np.array([0, 1, 2, 4], dtype=dt)])
    assert h.l_target == 2
    got = h.decompose(x_even.astype(dt))
    assert approx(got, expected), (got, expected)
    assert approx(h.recompose(expected.astype(dt)), x_even)


# ---- the scalar quantizer's known answers (tests/src/test_LinearQuantizer.cpp:94-111) ------------
# MGARD-CPU quantizes round(x / quantum); MGARD-X copysign(0.5 + |x * (1 / quantizer)|, x)
# (LinearQuantization.hpp:203-207): the same integers whenever 1 / quantizer is exact, which holds
# for the reference's vectors (quantum 0.5; dequantization: quantum * n with quantum 1.25). With
# s = inf every level has the quantizer 2 tol / ((l_target + 1)(1 + 3^D)) (:499-507): a 1-D grid of
# 5 nodes (l_target = 2) and ABS tol = 6 quantum gives exactly `quantum`.
def test_quantizer_known_answers():
    Q = G["quantizer"]
    qz = Q["quantize"]
    dt = DT[qz["dtype"]]
    x = np.array(qz["x"], dtype=dt)
    assert x.size == 5
    h = oracle.Hierarchy((5,), dt)
    tol = dt(6 * qz["quantum"])
    assert np.all(h.quantizers(oracle.ABS, tol, dt(np.inf), dt(1), False) == dt(qz["quantum"]))
    q, oi, ov, n = h.quantize(x, oracle.ABS, tol, dt(np.inf), dt(1), prep_huffman=False)
    assert n == 0 and q.tolist() == qz["n"]
    q2, oi, ov, n = h.quantize(x, oracle.ABS, tol, dt(np.inf), dt(1), dict_size=64, prep_huffman=True)
    assert n == 0 and (q2 - 32).tolist() == qz["n"]
    dq = Q["dequantize"]
    dt = DT[dq["dtype"]]
    ns = np.array(dq["n"] + [0], dtype=np.int64)          # padded to the 5-node grid
    h = oracle.Hierarchy((5,), dt)
    tol = dt(6 * dq["quantum"])
    back = h.dequantize(ns, oracle.ABS, tol, dt(np.inf), dt(1), prep_huffman=False)
    assert back.tolist() == dq["x"] + [0.0]
