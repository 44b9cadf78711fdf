"""BASELINE.json configs[0]: the serial CPU code path mgard::compress on a 1-D grid, restated in
oracle/mgcpu_1d.c (MGARD-CPU, not MGARD-X: SURVEY.md section 9 lists how the two differ), pinned
by the reference's own vectors for exactly this code path and then run at the configuration's
size, 2^20 float64."""
import json
import os

import numpy as np
import pytest

import oracle

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))
CASES = {(c["kind"], c["name"]): c for c in G["cases"]}


def approx(a, e, eps=1e-4):
    a, e = np.asarray(a, np.float64), np.asarray(e, np.float64)
    return np.all(np.abs(a - e) <= eps * np.abs(e) + 1e-6)


def test_hierarchy_of_dyadic_and_other_sizes():
    # TensorMeshHierarchy.tpp:52-95: 2, 3, 5, ..., 2^k + 1 [, n]
    h = oracle.MgardCpu1D(17)
    assert h.L == 4 and [h.level_size(l) for l in range(5)] == [2, 3, 5, 9, 17]
    h = oracle.MgardCpu1D(1 << 20)
    assert h.L == 20 and h.level_size(19) == (1 << 19) + 1 and h.level_size(20) == 1 << 20
    h = oracle.MgardCpu1D(20)
    assert h.L == 5 and [h.level_size(l) for l in range(6)] == [2, 3, 5, 9, 17, 20]


def test_decomposition_goldens_1d():
    """tests/src/test_decompose.cpp:277-338: expecteds[L] for the first 2^L + 1 entries of u_."""
    c = CASES[("decomposition", "1D, dyadic, uniform")]
    checked = 0
    for L, expected in enumerate(c["expecteds"]):
        n = (1 << L) + 1
        h = oracle.MgardCpu1D(n, coords=np.arange(n, dtype=np.float64))
        assert h.L == L
        assert approx(h.decompose(c["u"][:n]), expected), L
        checked += 1
    assert checked == 6


def test_decomposition_nonuniform_golden_1d():
    c = CASES[("decomposition", "1D, dyadic, nonuniform")]
    h = oracle.MgardCpu1D(3, coords=c["coordinates"][0][0])
    assert approx(h.decompose(c["u"]), c["expected"])     # (6.125, -3.5, 2.375)


def test_recomposition_goldens_1d():
    c = CASES[("recomposition", "1D, dyadic, uniform")]
    for L, expected in enumerate(c["expecteds"]):
        n = (1 << L) + 1
        h = oracle.MgardCpu1D(n, coords=np.arange(n, dtype=np.float64))
        assert approx(h.recompose(c["u"][:n]), expected), L


def test_quantizer_known_answers_cpu_form():
    """tests/src/test_LinearQuantizer.cpp:94-111 with the division form of LinearQuantizer.tpp:25."""
    Q = G["quantizer"]
    h = oracle.MgardCpu1D(5)
    assert h.quantize(Q["quantize"]["x"], Q["quantize"]["quantum"]).tolist() == Q["quantize"]["n"]
    h4 = oracle.MgardCpu1D(4)
    assert h4.dequantize(Q["dequantize"]["n"], Q["dequantize"]["quantum"]).tolist() == Q["dequantize"]["x"]


@pytest.mark.parametrize("n", [33, 20, 1000, 4097])
def test_roundtrip_and_error_bound_small(n):
    rng = np.random.default_rng(n)
    x = np.sort(rng.random(n)) if n == 1000 else None
    h = oracle.MgardCpu1D(n, coords=x)
    u = np.sin(np.linspace(0, 9, n)) + 0.01 * rng.standard_normal(n)
    c = h.decompose(u)
    assert np.max(np.abs(h.recompose(c) - u)) < 1e-12
    tol = 1e-3 * np.max(np.abs(u))
    qm = h.quantum(tol)
    back = h.recompose(h.dequantize(h.quantize(c, qm), qm))
    assert np.max(np.abs(back - u)) <= tol


def test_config0_1d_2pow20_f64_serial_cpu():
    """configs[0]: 1-D 2^20 float64 uniform grid, s = inf, tolerance 1e-3 max|u| -- the plumbing
    configuration, on the CPU restatement of mgard::compress (no GPU)."""
    n = 1 << 20
    rng = np.random.default_rng(20260101)
    t = np.arange(n) / n
    u = np.sin(2 * np.pi * 5 * t) + 0.1 * np.sin(2 * np.pi * 50 * t) + 1e-3 * rng.uniform(-1, 1, n)
    h = oracle.MgardCpu1D(n)
    tol = 1e-3 * float(np.max(np.abs(u)))
    c = h.decompose(u)
    qm = h.quantum(tol)
    assert qm == 2 * tol / (21 * 4)
    q = h.quantize(c, qm)
    back = h.recompose(h.dequantize(q, qm))
    assert np.max(np.abs(back - u)) <= tol
    # the coefficients of the finest level are the 1e-3 noise: a few dozen quanta, never thousands
    assert np.median(np.abs(q)) < 100
