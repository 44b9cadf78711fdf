"""Structural properties of the CPU oracle that pin the parts of the MGARD-X algorithm the
reference's golden vectors do not reach (they are all dyadic): the even-size "ghost node" rule,
the quantizer formulas and the hierarchy arrays. Reference citations are in
oracle/mgx_oracle_impl.h."""
import itertools

import numpy as np
import pytest

import oracle

F32 = np.float32


def lerp(v0, v1, t):
    # Coefficient/GPKFunctor.h:21-23, evaluated in the dtype of the operands
    r = v0 + v0 * t * v0.dtype.type(-1)
    return r + t * v1


def test_level_shapes_512():
    h = oracle.Hierarchy((512, 512, 512), F32)
    assert h.l_target == 9
    assert [h.level_shape(l)[0] for l in range(10)] == [2, 3, 5, 9, 17, 33, 65, 129, 257, 512]
    h = oracle.Hierarchy((17, 20, 33), F32)
    assert h.l_target == 4
    assert [h.level_shape(l) for l in range(5)] == [(2, 3, 3), (3, 4, 5), (5, 6, 9), (9, 11, 17),
                                                    (17, 20, 33)]
    with pytest.raises(ValueError):
        oracle.Hierarchy((2, 5), F32)


def test_hierarchy_arrays_even():
    # Hierarchy.hpp:29-42: even n splits the last cell in two halves; ratio there is 1/2
    h = oracle.Hierarchy((8,), np.float64, normalize_coordinates=False)
    assert h.level_shape(h.l_target) == (8,) and h.level_shape(h.l_target - 1) == (5,)
    np.testing.assert_array_equal(h.dist(h.l_target, 0), [1, 1, 1, 1, 1, 1, .5, .5])
    np.testing.assert_array_equal(h.dist(h.l_target - 1, 0), [2, 2, 2, 1, 0])
    assert h.ratio(h.l_target, 0)[6] == 0.5
    # Thomas coefficients: bm[0] = 1, am[n] = 0 (Hierarchy.hpp:146-155)
    assert h.bm(h.l_target, 0)[0] == 1 and h.am(h.l_target, 0)[8] == 0
    np.testing.assert_array_equal(h.marks(0), [0, 0, 1, 2, 2, 3, 3, 3])


def _embed_even(u, shape):
    """Insert a virtual mid-cell node in every even-sized dim, valued so that its multilevel
    coefficient is exactly zero (the interpolant of its coarse neighbours, f then c then r)."""
    D = len(shape)
    big_shape = tuple(n + 1 if n % 2 == 0 else n for n in shape)
    big = np.zeros(big_shape, dtype=u.dtype)
    # real node i -> i, except the last node of an even dim moves from n-1 to n
    maps = [np.array([i if (n % 2 or i < n - 1) else n for i in range(n)]) for n in shape]
    big[np.ix_(*maps)] = u
    virt = [n - 1 if n % 2 == 0 else None for n in shape]
    half = u.dtype.type(0.5)

    def interp(idx):
        # The node's interpolant from its even-index (coarse) neighbours: a lerp along every
        # dim in which the node's index is odd, last dim innermost (f, then c, then r). All the
        # ratios involved are 1/2 with these coordinates.
        def ev(cur, d):
            if d == D:
                return big[tuple(cur)]
            if cur[d] % 2 == 1:
                lo, hi = list(cur), list(cur)
                lo[d] -= 1
                hi[d] += 1
                return lerp(ev(lo, d + 1), ev(hi, d + 1), half)
            return ev(cur, d + 1)
        return ev(list(idx), 0)

    for idx in itertools.product(*[range(n) for n in big_shape]):
        if any(v is not None and idx[d] == v for d, v in enumerate(virt)):
            big[idx] = interp(idx)
    return big, big_shape


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(8,), (6, 9), (6, 8), (5, 6, 8), (6, 6, 6)])
def test_even_size_equals_odd_with_virtual_midpoint(shape, dt):
    """MGARD-X treats an even dim of size n as the odd dim n+1 whose extra node sits mid-way in
    the last cell and has a zero coefficient (SingleDimension/Coefficient/CoefficientKernel.hpp
    :87-105, Hierarchy.hpp:38-42, LinearProcessingKernel3D.hpp:52,177-203). With integer
    coordinates the half-cell spacings are exact, so both decompositions must agree BIT-exactly
    on every real node; the virtual nodes' coefficients must be exactly 0."""
    rng = np.random.default_rng(7)
    u = rng.standard_normal(shape).astype(dt)
    h = oracle.Hierarchy(shape, dt, normalize_coordinates=False)
    got = h.decompose(u)

    big, big_shape = _embed_even(u, shape)
    coords = []
    for n in shape:
        x = np.arange(n, dtype=dt)
        if n % 2 == 0:
            x = np.concatenate([x[:-1], [n - 1.5], x[-1:]]).astype(dt)
        coords.append(x)
    hb = oracle.Hierarchy(big_shape, dt, coords=coords)
    ref = hb.decompose(big)
    L = h.l_target
    assert hb.l_target == L
    # drop the virtual coefficient slot: last index of the finest-level coefficient block
    keep = [np.array([i for i in range(nb) if not (n % 2 == 0 and i == nb - 1)])
            for n, nb in zip(shape, big_shape)]
    sub = ref[np.ix_(*keep)]
    np.testing.assert_array_equal(got, sub)
    # and what was dropped is exactly zero
    mask = np.ones(big_shape, dtype=bool)
    mask[np.ix_(*keep)] = False
    assert np.all(ref[mask] == 0)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_coefficients_of_multilinear_function_vanish(dt):
    # mirrors the reference's "coefficients of linear functions" test (test_decompose.cpp:477):
    # a function that is multilinear has (near-)zero multilevel coefficients on every level
    shape = (9, 12, 17)
    x = [np.linspace(0, 1, n) for n in shape]
    u = (1 + 2 * x[0][:, None, None]) * (3 - x[1][None, :, None]) * (0.5 + x[2][None, None, :])
    u = u.astype(dt)
    h = oracle.Hierarchy(shape, dt)
    c = h.decompose(u)
    cs = h.level_shape(0)
    mask = np.ones(shape, dtype=bool)
    mask[:cs[0], :cs[1], :cs[2]] = False
    assert np.max(np.abs(c[mask])) < (1e-5 if dt == np.float32 else 1e-13)


def test_decompose_is_linear():
    shape = (10, 9, 12)
    rng = np.random.default_rng(5)
    a = rng.standard_normal(shape)
    b = rng.standard_normal(shape)
    h = oracle.Hierarchy(shape, np.float64)
    lhs = h.decompose(2.5 * a + b)
    rhs = 2.5 * h.decompose(a) + h.decompose(b)
    assert np.max(np.abs(lhs - rhs)) < 1e-12


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_quantizer_known_answers(dt):
    # LinearQuantization.hpp:495-545
    h = oracle.Hierarchy((9, 9, 9), dt)  # l_target = 3, D = 3
    tol, norm = dt(1e-2), dt(4.0)
    inf = dt(np.inf)
    q = h.quantizers(oracle.REL, tol, inf, norm, False)
    expect = dt(2.0 * float(tol) * float(norm) / (4 * 28.0))
    assert np.all(q == expect)
    qr = h.quantizers(oracle.REL, tol, inf, norm, True)
    assert np.all(qr == dt(1.0) / expect)
    q_abs = h.quantizers(oracle.ABS, tol, inf, norm, False)
    assert np.all(q_abs == dt(2.0 * float(tol) / (4 * 28.0)))
    q0 = h.quantizers(oracle.ABS, tol, dt(0), norm, False)
    assert np.all(q0 == dt(2.0 * float(tol) / np.sqrt(729.0)))
    q1 = h.quantizers(oracle.ABS, tol, dt(1), norm, False)
    for l in range(4):
        assert q1[l] == dt(2.0 * float(tol) / (2.0 ** l * np.sqrt(729.0)))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_quantize_rule_and_outliers(dt):
    # LinearQuantization.hpp:203-245: q = trunc(copysign(0.5+|t*quantizer|, t)) + dict/2, outliers
    h = oracle.Hierarchy((5,), dt)  # l_target = 2, D = 1 -> quantizer = 2*tol/(3*4)
    tol = dt(0.6)
    binw = dt(2.0 * float(tol) / 12.0)
    rq = dt(1.0) / binw
    v = np.array([0.0, 0.04, -0.06, 0.26, -1000.0], dtype=dt)
    q, oi, ov, cnt = h.quantize(v, oracle.ABS, tol, dt(np.inf), dt(1), dict_size=64)
    exp = [int(np.trunc(np.copysign(dt(0.5) + abs(t * rq), t))) for t in v]
    assert cnt == 1 and list(oi) == [4] and list(ov) == [exp[4] + 32]
    assert list(q) == [exp[0] + 32, exp[1] + 32, exp[2] + 32, exp[3] + 32, 0]
    back = h.dequantize(q, oracle.ABS, tol, dt(np.inf), dt(1), dict_size=64, outlier_idx=oi,
                        outlier_val=ov)
    assert np.all(np.abs(back - v) <= binw / 2 * (1 + 1e-6))


@pytest.mark.parametrize("s", [np.inf, 0.0])
@pytest.mark.parametrize("shape", [(33,), (17, 20), (12, 9, 18)])
def test_roundtrip_error_bound(shape, s):
    """compress -> decompress keeps the error within tolerance (the reference's own
    property test, tests/src/test_compress.cpp:75-118, here for the mgard_x path)."""
    rng = np.random.default_rng(11)
    x = [np.linspace(0, 1, n) for n in shape]
    u = np.zeros(shape)
    for d, xd in enumerate(x):
        sh = [1] * len(shape)
        sh[d] = -1
        u = u + np.sin(2 * np.pi * (d + 1) * xd).reshape(sh)
    u = (u + 0.01 * rng.standard_normal(shape)).astype(np.float64)
    h = oracle.Hierarchy(shape, np.float64)
    tol = 1e-3
    nrm = oracle.norm(u, s)
    c = h.decompose(u)
    q, oi, ov, cnt = h.quantize(c, oracle.REL, tol, s, nrm)
    back = h.recompose(h.dequantize(q, oracle.REL, tol, s, nrm, outlier_idx=oi, outlier_val=ov))
    if np.isinf(s):
        assert np.max(np.abs(back - u)) <= tol * nrm
    else:
        err = np.sqrt(np.mean((back - u) ** 2))
        assert err <= tol * nrm


def test_level_linearise_hand_derived_3x3_and_bijection():
    """config.reorder == 1 (LinearQuantization.hpp:46-146, 588-605), restated in the oracle.
    3 x 3 grid, one level: the four level-0 nodes come first (row-major in the 2 x 2 grid), then
    the five level-1 coefficients in the row-major order of their NATURAL positions
    (0,1) (1,0) (1,1) (1,2) (2,1). In the reordered layout (coarse index first along each
    dim: natural 0, 2, 1) those sit at (0,2) (2,0) (2,2) (2,1) (1,2)."""
    import numpy as np
    import oracle
    o = oracle.Hierarchy((3, 3), np.float32)
    q = np.arange(9, dtype=np.int64).reshape(3, 3)      # value = reordered linear index
    lin = o.level_linearize(q)
    want = [0 * 3 + 0, 0 * 3 + 1, 1 * 3 + 0, 1 * 3 + 1,            # level 0
            0 * 3 + 2, 2 * 3 + 0, 2 * 3 + 2, 2 * 3 + 1, 1 * 3 + 2]  # level 1
    assert lin.tolist() == want
    for shape in [(5,), (8,), (9, 6), (5, 9, 17), (6, 8, 10), (5, 5, 5, 5), (3, 4, 5, 6, 7), (33, 20, 17)]:
        h = oracle.Hierarchy(shape, np.float64)
        n = int(np.prod(shape))
        a = np.arange(n, dtype=np.int64)
        f = h.level_linearize(a)
        assert np.array_equal(np.sort(f), a)                       # a permutation
        assert np.array_equal(h.level_linearize(f, inverse=True).reshape(-1), a)
        # level by level: the slot of level l holds exactly the elements whose level is l
        pos = np.array([h.linearized_position(i) for i in range(n)])
        sizes = [int(np.prod(h.level_shape(l))) for l in range(h.l_target + 1)]
        marks = [np.asarray(h.marks(d)) for d in range(len(shape))]
        idx = np.unravel_index(np.arange(n), shape)
        lev = np.max([marks[d][idx[d]] for d in range(len(shape))], axis=0)
        for l in range(h.l_target + 1):
            lo = sizes[l - 1] if l else 0
            assert np.all((pos[lev == l] >= lo) & (pos[lev == l] < sizes[l]))
