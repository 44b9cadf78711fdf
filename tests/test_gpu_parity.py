"""GPU parity tests: the HIP path (through the C ABI, libmgard_hip.so) against the CPU oracle on
the same seeded inputs. Bars (BASELINE.json north_star): quantized integers bit-exact, float
multilevel coefficients within 1 ULP -- the checks below demand bit-exact floats too, which the
design achieves by keeping the reference's per-element operation order with FMA contraction off.
"""
import json
import os

import numpy as np
import pytest

import oracle
from tests.util import nonuniform_coords, smooth_field

pytestmark = pytest.mark.gpu

SHAPES = [(5,), (33,), (6,), (100,), (1025,), (5, 5), (17, 20), (6, 9), (64, 48), (129, 33),
          (5, 5, 5), (9, 6, 8), (17, 20, 33), (33, 33, 33), (34, 33, 32), (65, 70, 129),
          (3, 3, 3), (4, 4, 4), (3, 4, 200),
          # (short fastest extent under a long middle one: the 64 x 4 tiles of round 6)
          (40, 130, 9), (33, 100, 17), (36, 128, 12), (20, 200, 30), (70, 300, 5)]


def _gpu():
    import torch
    import mgard_amd
    return torch, mgard_amd


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def assert_bit_equal(got, ref, what=""):
    gb, rb = _bits(got), _bits(ref)
    if not np.array_equal(gb, rb):
        bad = np.argwhere(gb != rb)
        i = tuple(bad[0])
        raise AssertionError("%s: %d/%d elements differ; first at %s: got %r ref %r" % (
            what, len(bad), gb.size, i, got[i], ref[i]))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(512,), (17, 20), (34, 33, 32)])
def test_hierarchy_tables_match_oracle(shape, dt):
    torch, mg = _gpu()
    coords = nonuniform_coords(shape, dt)
    for cs in (None, coords):
        h = mg.Hierarchy(shape, dt, coords=cs)
        o = oracle.Hierarchy(shape, dt, coords=cs)
        assert h.l_target == o.l_target
        for l in range(h.l_target + 1):
            assert h.level_shape(l) == o.level_shape(l)
            for d in range(len(shape)):
                for kind in ("dist", "ratio", "am", "bm"):
                    assert_bit_equal(h.table(kind, l, d), getattr(o, kind)(l, d), kind)
        for d in range(len(shape)):
            np.testing.assert_array_equal(h.table("marks", h.l_target, d), o.marks(d))
        h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_decompose_recompose_bit_exact(shape, dt):
    torch, mg = _gpu()
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    c = h.decompose(ud)
    ref = o.decompose(u)
    assert_bit_equal(c.cpu().numpy(), ref, "decompose %r" % (shape,))
    assert torch.equal(ud.cpu(), torch.from_numpy(u)), "input must not be modified"
    # in-place variant (the reference's Decompose is in place)
    ud2 = ud.clone()
    h.decompose(ud2, out=ud2)
    assert_bit_equal(ud2.cpu().numpy(), ref, "decompose in place")
    # recompose
    back = h.recompose(c)
    assert_bit_equal(back.cpu().numpy(), o.recompose(ref), "recompose %r" % (shape,))
    c2 = c.clone()
    h.recompose(c2, out=c2)
    assert torch.equal(c2, back)
    h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(100,), (17, 20), (9, 6, 8), (34, 33, 32), (65, 70, 129)])
def test_decompose_nonuniform_bit_exact(shape, dt):
    torch, mg = _gpu()
    coords = nonuniform_coords(shape, dt)
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt, coords=coords)
    o = oracle.Hierarchy(shape, dt, coords=coords)
    c = h.decompose(torch.from_numpy(u).cuda())
    ref = o.decompose(u)
    assert_bit_equal(c.cpu().numpy(), ref, "nonuniform decompose")
    assert_bit_equal(h.recompose(c).cpu().numpy(), o.recompose(ref), "nonuniform recompose")
    h.close()


def _outlier_set(idx, val):
    idx = np.asarray(idx).astype(np.int64)
    val = np.asarray(val).astype(np.int64)
    order = np.argsort(idx, kind="stable")
    return idx[order], val[order]


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("s", [np.inf, 0.0, 1.0, -1.0])
@pytest.mark.parametrize("mode", ["REL", "ABS"])
@pytest.mark.parametrize("shape", [(100,), (17, 20), (34, 33, 32)])
def test_quantize_dequantize_bit_exact(shape, mode, s, dt):
    torch, mg = _gpu()
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    coef = o.decompose(u)
    eb = oracle.REL if mode == "REL" else oracle.ABS
    tol = 1e-3
    # the L2 norm is not bit-reproducible across implementations, so the
    # oracle's norm is injected, as the low-level reference API allows (Compressor.h:73-78)
    nrm = oracle.norm(u, dt(s))
    for dict_size in (8192, 64):
        q, oi, ov, n = h.quantize(torch.from_numpy(coef).cuda(), eb, tol, s, nrm, dict_size=dict_size)
        rq, roi, rov, rn = o.quantize(coef, eb, dt(tol), dt(s), dt(nrm), dict_size=dict_size)
        assert n == rn
        np.testing.assert_array_equal(q.cpu().numpy(), rq)
        gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
        ri, rv = _outlier_set(roi, rov)
        np.testing.assert_array_equal(gi, ri)
        np.testing.assert_array_equal(gv, rv)
        back = h.dequantize(q.clone(), eb, tol, s, nrm, dict_size=dict_size, outlier_idx=oi,
                            outlier_val=ov)
        rback = o.dequantize(rq, eb, dt(tol), dt(s), dt(nrm), dict_size=dict_size,
                             outlier_idx=roi, outlier_val=rov)
        assert_bit_equal(back.cpu().numpy(), rback, "dequantize")
    # without the Huffman shift (lossless == CPU_Lossless in the reference)
    q, oi, ov, n = h.quantize(torch.from_numpy(coef).cuda(), eb, tol, s, nrm, prep_huffman=False)
    rq, _, _, _ = o.quantize(coef, eb, dt(tol), dt(s), dt(nrm), prep_huffman=False)
    assert n == 0
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_norm(dt):
    torch, mg = _gpu()
    shape = (65, 70, 129)
    u = smooth_field(shape, dt)
    h = mg.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    assert h.norm(ud, np.inf) == oracle.norm(u, dt(np.inf))  # max|x| is order independent
    # L2: the oracle accumulates sequentially in T like the reference's SERIAL backend
    # (DeviceAdapterSerial.h:1376-1381), the GPU reduces as a tree; neither is bit-reproducible
    #  -- compare both against a float64 evaluation instead.
    l2 = h.norm(ud, 0.0)
    exact = float(np.sqrt(np.mean(u.astype(np.float64) ** 2)))
    assert abs(l2 - exact) <= (1e-6 if dt == np.float32 else 1e-14) * exact
    assert abs(oracle.norm(u, dt(0)) - exact) <= (5e-4 if dt == np.float32 else 1e-12) * exact
    z = torch.zeros_like(ud)
    assert h.norm(z, np.inf) == np.finfo(dt).eps  # NormCalculator.hpp:50-52
    h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1025,), (129, 33), (34, 33, 32), (65, 70, 129)])
def test_fused_decompose_quantize_equals_stages(shape, dt):
    torch, mg = _gpu()
    u = smooth_field(shape, dt)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    q, oi, ov, n, nrm = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf)
    assert nrm == oracle.norm(u, dt(np.inf))
    rq, roi, rov, rn = o.quantize(o.decompose(u), oracle.REL, dt(1e-3), dt(np.inf), dt(nrm))
    assert n == rn
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gv, rv)
    # and the way back: error bound (north_star: round-trip L-inf error <= requested tolerance)
    back = h.dequantize_recompose(q, mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    err = np.max(np.abs(back.cpu().numpy().astype(np.float64) - u.astype(np.float64)))
    assert err <= 1e-3 * nrm
    h.close()


def test_reference_goldens_through_gpu():
    """The reference's own decomposition goldens (tests/src/test_decompose.cpp) through the HIP
    path, tolerance = the reference test's Catch::Approx epsilon 1e-4."""
    torch, mg = _gpu()
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))
    dts = {"float": np.float32, "double": np.float64}
    n_checked = 0
    for c in G["cases"]:
        if c["ndim"] > 3 or "expecteds" not in c:
            continue
        dt = dts[c["dtype"]]
        for L, expected in enumerate(c["expecteds"]):
            n = (1 << L) + 1
            if n < 3:
                continue
            shape = (n,) * c["ndim"]
            u = np.array(c["u"][: n ** c["ndim"]], dtype=dt).reshape(shape)
            h = mg.Hierarchy(shape, dt, normalize_coordinates=False)
            if c["kind"] == "decomposition":
                got = oracle.dyadic_reordered_to_natural(h.decompose(torch.from_numpy(u).cuda()).cpu().numpy())
            else:
                got = h.recompose(torch.from_numpy(oracle.dyadic_natural_to_reordered(u)).cuda()).cpu().numpy()
            exp = np.array(expected, dtype=np.float64)
            assert np.all(np.abs(got.ravel() - exp) <= 1e-4 * np.abs(exp) + 1e-5), (c["name"], L)
            h.close()
            n_checked += 1
    assert n_checked >= 9


def test_invalid_arguments():
    torch, mg = _gpu()
    with pytest.raises(mg.MgardHipError):
        mg.Hierarchy((2, 5))          # dims < 3 are rejected (Hierarchy.hpp:742-756)
    with pytest.raises(mg.MgardHipError):
        mg.Hierarchy((5,) * 6)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(17, 20, 33), (34, 33, 32), (65, 70, 129)])
def test_simple_kernels_match_fused_kernels(shape, dt, monkeypatch):
    """MGH_FORCE_V1=1 selects the one-thread-per-element kernels; both kernel sets must give
    the oracle's bits."""
    torch, mg = _gpu()
    u = smooth_field(shape, dt, noise=1e-2)
    ref = oracle.Hierarchy(shape, dt).decompose(u)
    for flag in ("1", "0"):
        monkeypatch.setenv("MGH_FORCE_V1", flag)
        h = mg.Hierarchy(shape, dt)
        c = h.decompose(torch.from_numpy(u).cuda())
        assert_bit_equal(c.cpu().numpy(), ref, "MGH_FORCE_V1=%s" % flag)
        q, oi, ov, n, nrm = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.ABS, 1e-3, np.inf, 1.0)
        rq, roi, rov, rn = oracle.Hierarchy(shape, dt).quantize(ref, oracle.ABS, dt(1e-3), dt(np.inf), dt(1))
        assert n == rn
        np.testing.assert_array_equal(q.cpu().numpy(), rq)
        h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("s,nsub", [(np.inf, 8), (0.0, 4)])
def test_device_norm_path_decomposed_domain(dt, s, nsub):
    """mgh_norm_device + mgh_decompose_quantize_dn (nothing returns to the host) apply the
    reference's per-subdomain bound calc_local_abs_tol (ErrorToleranceCalculator.hpp:134-155)."""
    torch, mg = _gpu()
    shape = (34, 33, 65)
    u = smooth_field(shape, dt)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    dn = h.norm_device(ud, s)
    g = dt(dn.item())
    if np.isinf(s):
        assert g == dt(oracle.norm(u, dt(s)))
        g_used = g
        atol = dt(1e-3) * g_used
    else:
        g_used = dt(oracle.norm(u, dt(s)))  # L2 is not bit-reproducible: inject the oracle's
        dn.fill_(float(g_used))
        a = (dt(1e-3) * g_used) * (dt(1e-3) * g_used) / dt(nsub)
        atol = np.sqrt(a, dtype=dt)
    cap = u.size
    bufs = (torch.empty(shape, dtype=torch.int64, device="cuda"),
            torch.zeros(1, dtype=torch.int64, device="cuda"),
            torch.empty(cap, dtype=torch.int64, device="cuda"),
            torch.empty(cap, dtype=torch.int64, device="cuda"))
    q, oi, ov, cnt = h.decompose_quantize_dn(ud, mg.REL, 1e-3, s, dn, nsub, bufs)
    rq, roi, rov, rn = o.quantize(o.decompose(u), oracle.ABS, dt(atol), dt(s), dt(1))
    n = int(cnt.item())
    assert n == rn
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi[:n].cpu().numpy(), ov[:n].cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gv, rv)
    h.close()


def _host_mem_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


@pytest.mark.parametrize("n", [512, 1024])
def test_full_size_roundtrip_and_parity(n):
    """BASELINE.json configs[1] (512^3) and configs[4] (1024^3 compress + decompress round trip):
    quantized integers bit-exact vs the CPU oracle at FULL size, and the round-trip L-inf error
    within the requested tolerance."""
    torch, mg = _gpu()
    shape = (n, n, n)
    if _host_mem_gb() < (24 if n == 512 else 120):
        pytest.skip("not enough host memory for the full-size oracle run")
    u = smooth_field(shape, np.float32)
    h = mg.Hierarchy(shape, np.float32)
    ud = torch.from_numpy(u).cuda()
    tol = 1e-3
    q, oi, ov, cnt, nrm = h.decompose_quantize(ud, mg.REL, tol, np.inf, outlier_cap=u.size // 8)
    assert cnt <= u.size // 8
    assert nrm == float(np.max(np.abs(u)))
    back = h.dequantize_recompose(q.clone(), mg.REL, tol, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    err = float((back - ud).abs().max().item())
    assert err <= tol * nrm, (err, tol * nrm)
    del back
    o = oracle.Hierarchy(shape, np.float32)
    c = o.decompose(u)
    rq, roi, rov, rn = o.quantize(c, oracle.REL, np.float32(tol), np.float32(np.inf), np.float32(nrm),
                                  outlier_cap=u.size // 8)
    del c
    assert cnt == rn
    assert np.array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    assert np.array_equal(gi, ri) and np.array_equal(gv, rv)
    h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("s", [np.inf, 0.0])
@pytest.mark.parametrize("shape", [(5, 5, 5), (17, 20, 33), (34, 33, 32), (65, 70, 129), (100, 36, 260)])
def test_fused_dequantize_recompose_bit_exact(shape, s, dt):
    """mgh_dequantize_recompose (fused kernels for D = 3) against the oracle's
    dequantize + recompose, with outliers and a small dictionary."""
    torch, mg = _gpu()
    u = smooth_field(shape, dt, noise=1e-2)
    o = oracle.Hierarchy(shape, dt)
    nrm = oracle.norm(u, dt(s))
    rq, roi, rov, rn = o.quantize(o.decompose(u), oracle.REL, dt(1e-3), dt(s), dt(nrm), dict_size=256)
    ref = o.recompose(o.dequantize(rq, oracle.REL, dt(1e-3), dt(s), dt(nrm), dict_size=256,
                                   outlier_idx=roi, outlier_val=rov))
    h = mg.Hierarchy(shape, dt)
    back = h.dequantize_recompose(torch.from_numpy(rq).cuda(), mg.REL, 1e-3, s, nrm, dict_size=256,
                                  outlier_idx=torch.from_numpy(roi.astype(np.int64)).cuda(),
                                  outlier_val=torch.from_numpy(rov).cuda())
    assert_bit_equal(back.cpu().numpy(), ref, "fused dequantize+recompose %r" % (shape,))
    h.close()


@pytest.mark.parametrize("nd_ipk", [None, "1"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(5, 5, 5, 5), (3, 3, 3, 3), (6, 9, 8, 12), (17, 5, 20, 33), (3, 4, 5, 6, 7),
                                   (5, 5, 9, 6, 10), (3, 3, 3, 5, 6000), (3, 2600, 3, 4, 5)])
def test_nd_decompose_recompose_bit_exact(shape, dt, nd_ipk, monkeypatch):
    """D = 4, 5: generic N-D kernels against the oracle's N-D
    restatement, which itself reproduces the reference's 4-D golden vectors and equals the
    3-D code on D <= 3. The Thomas solves of the N-D path run on the 3-D kernels through a
    (dims before, dim, dims behind) view of the compact box -- a long dimension in verified
    chunks -- or (MGH_ND_IPK=1) on the path's own one-thread-per-pencil kernel."""
    torch, mg = _gpu()
    if nd_ipk:
        monkeypatch.setenv("MGH_ND_IPK", nd_ipk)
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    assert h.l_target == o.l_target
    ud = torch.from_numpy(u).cuda()
    c = h.decompose(ud)
    ref = o.decompose(u)
    assert_bit_equal(c.cpu().numpy(), ref, "nd decompose %r" % (shape,))
    back = h.recompose(c)
    assert_bit_equal(back.cpu().numpy(), o.recompose(ref), "nd recompose %r" % (shape,))
    # quantized path (staged for D > 3) and round-trip bound
    nrm = oracle.norm(u, dt(np.inf))
    q, oi, ov, n, nrm2 = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf)
    assert nrm2 == nrm
    rq, roi, rov, rn = o.quantize(ref, oracle.REL, dt(1e-3), dt(np.inf), dt(nrm))
    assert n == rn
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    rt = h.dequantize_recompose(q, mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    assert float((rt - ud).abs().max().item()) <= 1e-3 * nrm
    h.close()


@pytest.mark.parametrize("shape", [(33,), (17, 20), (9, 6, 8), (34, 33, 32)])
def test_nd_kernels_equal_3d_kernels(shape, monkeypatch):
    """MGH_FORCE_ND=1 runs the generic N-D kernels on D <= 3 inputs: same bits as the oracle."""
    torch, mg = _gpu()
    u = smooth_field(shape, np.float32, noise=1e-2)
    o = oracle.Hierarchy(shape, np.float32)
    ref = o.decompose(u)
    monkeypatch.setenv("MGH_FORCE_ND", "1")
    h = mg.Hierarchy(shape, np.float32)
    c = h.decompose(torch.from_numpy(u).cuda())
    assert_bit_equal(c.cpu().numpy(), ref, "forced nd decompose")
    assert_bit_equal(h.recompose(c).cpu().numpy(), o.recompose(ref), "forced nd recompose")
    h.close()


@pytest.mark.parametrize("rows", ["1", "0"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(6, 9, 8, 12), (3, 4, 5, 6, 7), (5, 5, 9, 6, 10), (4, 3, 70, 5, 131),
                                   (8, 8, 20, 20, 20), (17, 20), (9, 6, 8), (3, 3, 3, 3, 200),
                                   (3, 3, 4, 18, 66), (3, 4, 5, 3, 9)])
def test_generic_nd_row_kernels_bit_exact(shape, dt, rows, monkeypatch):
    """The generic N-D path on its row-wise kernels (round 6: a wave per group of rows of one line of
    dimension 3, corner rows loaded once and traded between lanes, rows longer than the wave 62
    elements a round; the sweeps on 3-D views; MGH_ND_ROWS=0: one thread per element) --
    forced for D <= 4 as well (MGH_FORCE_ND / MGH_FUSED4=0) -- against the oracle, out of place
    (the top level reads the input where it is) and in place, plus the quantizer behind it
    (slot requests by ballot) with a dictionary small enough to leave outliers."""
    torch, mg = _gpu()
    monkeypatch.setenv("MGH_ND_ROWS", rows)
    monkeypatch.setenv("MGH_FUSED4", "0")
    if len(shape) <= 3:
        monkeypatch.setenv("MGH_FORCE_ND", "1")
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    ref = o.decompose(u)
    c = h.decompose(ud)
    assert_bit_equal(c.cpu().numpy(), ref, "nd decompose %r" % (shape,))
    assert torch.equal(ud.cpu(), torch.from_numpy(u)), "the input of an out-of-place call was modified"
    w = ud.clone()
    h.decompose(w, out=w)
    assert_bit_equal(w.cpu().numpy(), ref, "nd decompose in place %r" % (shape,))
    back = h.recompose(c)
    assert_bit_equal(back.cpu().numpy(), o.recompose(ref), "nd recompose %r" % (shape,))
    nrm = oracle.norm(u, dt(np.inf))
    q, oi, ov, n, nrm2 = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf, dict_size=16)
    rq, roi, rov, rn = o.quantize(ref, oracle.REL, dt(1e-3), dt(np.inf), dt(nrm), dict_size=16)
    assert n == rn and rn > 0
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    assert sorted(zip(oi.tolist(), ov.tolist())) == sorted(zip(roi.tolist(), rov.tolist()))
    h.close()


def test_reference_4d_goldens_through_gpu():
    torch, mg = _gpu()
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))
    n_checked = 0
    for c in G["cases"]:
        if c["ndim"] != 4:
            continue
        shape = (3, 3, 3, 3)
        u = np.array(c["u"][:81], dtype=np.float32).reshape(shape)
        h = mg.Hierarchy(shape, np.float32, normalize_coordinates=False)
        if c["kind"] == "decomposition":
            got = oracle.dyadic_reordered_to_natural(h.decompose(torch.from_numpy(u).cuda()).cpu().numpy())
        else:
            got = h.recompose(torch.from_numpy(oracle.dyadic_natural_to_reordered(u)).cuda()).cpu().numpy()
        exp = np.array(c["expecteds"][1], dtype=np.float64)
        assert np.all(np.abs(got.ravel() - exp) <= 1e-4 * np.abs(exp) + 1e-5), c["kind"]
        h.close()
        n_checked += 1
    assert n_checked == 2


def _random_cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    cases = []
    budget = {1: 5000, 2: 40000, 3: 150000, 4: 150000, 5: 120000}
    while len(cases) < n_cases:
        D = int(rng.integers(1, 6))
        lo, hi = 3, {1: 3000, 2: 220, 3: 70, 4: 22, 5: 12}[D]
        shape = tuple(int(x) for x in rng.integers(lo, hi + 1, size=D))
        if np.prod(shape) > budget[D]:
            continue
        cases.append((shape, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                      [np.inf, 0.0, 1.0][int(rng.integers(0, 3))], bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize("case", _random_cases(48, 20261001), ids=lambda c: "x".join(map(str, c[0])))
def test_random_shapes_bit_exact(case):
    """Seeded sweep over ragged shapes in 1-5 D (even/odd/mixed sizes, levels limited by the
    smallest dim), f32/f64, uniform/non-uniform grids, s in {inf, 0, 1}, REL/ABS: decompose,
    quantize (+outliers) and the way back, all bit-exact against the oracle."""
    torch, mg = _gpu()
    shape, f64, nonuni, s, rel = case
    dt = np.float64 if f64 else np.float32
    coords = nonuniform_coords(shape, dt, seed=sum(shape)) if nonuni else None
    u = smooth_field(shape, dt, seed=int(np.prod(shape)), noise=1e-2)
    h = mg.Hierarchy(shape, dt, coords=coords)
    o = oracle.Hierarchy(shape, dt, coords=coords)
    ud = torch.from_numpy(u).cuda()
    ref = o.decompose(u)
    c = h.decompose(ud)
    assert_bit_equal(c.cpu().numpy(), ref, "decompose")
    eb = oracle.REL if rel else oracle.ABS
    nrm = oracle.norm(u, dt(s)) if rel else 1.0
    tol = 1e-3 if rel else 1e-3 * float(np.max(np.abs(u)))
    q, oi, ov, n, _ = h.decompose_quantize(ud, eb, tol, s, nrm, dict_size=512)
    rq, roi, rov, rn = o.quantize(ref, eb, dt(tol), dt(s), dt(nrm), dict_size=512)
    assert n == rn
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gv, rv)
    back = h.dequantize_recompose(q, eb, tol, s, nrm, dict_size=512, outlier_idx=oi, outlier_val=ov)
    rback = o.recompose(o.dequantize(rq, eb, dt(tol), dt(s), dt(nrm), dict_size=512,
                                     outlier_idx=roi, outlier_val=rov))
    assert_bit_equal(back.cpu().numpy(), rback, "dequantize+recompose")
    h.close()


def _compare_quantized(mg, h, o, u, ud, mode, tol, s, norm, cap):
    """decompose+quantize on the GPU (stages fused where the library fuses them) vs the oracle."""
    dt = u.dtype.type
    q, oi, ov, cnt, _ = h.decompose_quantize(ud, mode, tol, s, norm=float(norm), outlier_cap=cap)
    c = o.decompose(u)
    rq, roi, rov, rn = o.quantize(c, mode, dt(tol), dt(s), dt(norm), outlier_cap=cap)
    del c
    assert cnt == rn and cnt <= cap
    assert np.array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    assert np.array_equal(gi, ri) and np.array_equal(gv, rv)
    return q, oi, ov


def test_config0_1d_2pow20_f64():
    """BASELINE.json configs[0] shape and type (1-D 2^20 float64, uniform grid, s=inf, tolerance
    1e-3*maxabs) through the MGARD-X path: an even, non-dyadic size, so every level applies the
    ghost-node rule. Bit-exact integers vs the oracle + round trip within the tolerance."""
    torch, mg = _gpu()
    n = 1 << 20
    rng = np.random.default_rng(20260101)
    i = np.arange(n) / n
    u = np.sin(2 * np.pi * 5 * i) + 0.1 * np.sin(2 * np.pi * 50 * i) + 1e-3 * rng.uniform(-1, 1, n)
    h = mg.Hierarchy((n,), np.float64)
    o = oracle.Hierarchy((n,), np.float64)
    ud = torch.from_numpy(u).cuda()
    tol = 1e-3 * float(np.max(np.abs(u)))
    q, oi, ov = _compare_quantized(mg, h, o, u, ud, mg.ABS, tol, np.inf, 1.0, n // 4)
    back = h.dequantize_recompose(q.clone(), mg.ABS, tol, np.inf, 1.0, outlier_idx=oi, outlier_val=ov)
    assert float((back - ud).abs().max().item()) <= tol
    h.close()


def test_config2_512cube_f64_nonuniform_s0():
    """BASELINE.json configs[2] at full size: 512^3 float64, non-uniform coordinates, s=0
    (volume-weighted quantizers, non-constant mass matrix and Thomas coefficients). ABS mode as
    SURVEY.md section 9 caveat 1 prescribes (the L2 norm is not bit-reproducible)."""
    torch, mg = _gpu()
    shape = (512, 512, 512)
    if _host_mem_gb() < 40:
        pytest.skip("not enough host memory for the full-size oracle run")
    u = smooth_field(shape, np.float64)
    coords = nonuniform_coords(shape, np.float64)
    h = mg.Hierarchy(shape, np.float64, coords=coords)
    o = oracle.Hierarchy(shape, np.float64, coords=coords)
    ud = torch.from_numpy(u).cuda()
    _compare_quantized(mg, h, o, u, ud, mg.ABS, 1e-3, 0.0, 1.0, u.size // 8)
    h.close()


def test_config2_512cube_f64_nonuniform_s0_rel_roundtrip():
    """configs[2] as BASELINE.json states it -- REL, s = 0 -- at full size. The integers of a REL
    run depend on the L2 norm, which no two implementations sum in the same order, so this is
    the size-independent property instead: with the device's own norm (close to numpy's), the
    reconstruction error is within tol * norm in the root-mean-square sense the reference bounds
    for s = 0 (ErrorToleranceCalculator.hpp:69-89), everything on the device."""
    torch, mg = _gpu()
    shape = (512, 512, 512)
    u = smooth_field(shape, np.float64)
    coords = nonuniform_coords(shape, np.float64)
    h = mg.Hierarchy(shape, np.float64, coords=coords)
    ud = torch.from_numpy(u).cuda()
    tol = 1e-3
    q, oi, ov, cnt, nrm = h.decompose_quantize(ud, mg.REL, tol, 0.0, outlier_cap=u.size // 8)
    ref_norm = float(np.sqrt(np.mean(u ** 2)))
    assert cnt <= u.size // 8 and abs(nrm - ref_norm) <= 1e-9 * ref_norm
    back = h.dequantize_recompose(q, mg.REL, tol, 0.0, nrm, outlier_idx=oi, outlier_val=ov)
    err = float(torch.sqrt(torch.mean((back - ud) ** 2)).item())
    assert 0.0 < err <= tol * nrm
    h.close()


@pytest.mark.parametrize("shape", [(8, 128, 128, 128), (8, 64, 200, 96)])
def test_config3_4d_slab(shape):
    """BASELINE.json configs[3]: one rank's 4-D slab (8 x n^3 float32; the level count is limited
    by the short dimension, l_target = 3) at a size the oracle finishes in seconds; generic N-D
    kernels, bit-exact integers + round trip."""
    torch, mg = _gpu()
    u = smooth_field(shape, np.float32)
    h = mg.Hierarchy(shape, np.float32)
    o = oracle.Hierarchy(shape, np.float32)
    assert h.l_target == 3
    ud = torch.from_numpy(u).cuda()
    nrm = float(np.max(np.abs(u)))
    q, oi, ov = _compare_quantized(mg, h, o, u, ud, mg.REL, 1e-3, np.inf, nrm, u.size // 2)
    back = h.dequantize_recompose(q.clone(), mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    assert float((back - ud).abs().max().item()) <= 1e-3 * nrm
    h.close()


def _slab4d(shape, dt=np.float32):
    """A slab of time steps: a 3-D field that drifts slowly along dim 0 (tools/exp_4d.py)."""
    base = smooth_field(shape[1:], dt)
    return np.stack([base * dt(1.0 + 0.002 * t) + dt(1e-4 * t) for t in range(shape[0])])


@pytest.mark.parametrize("shape,dt", [((8, 66, 70, 129), np.float32), ((7, 33, 130, 65), np.float64),
                                      ((5, 40, 36, 72), np.float32), ((16, 65, 65, 65), np.float32),
                                      # (long r-pencils: the batched solve of the slices in verified chunks)
                                      ((3, 2100, 9, 10), np.float32), ((4, 4200, 6, 7), np.float64),
                                      # (64 x 4 tiles: a short fastest extent)
                                      ((6, 20, 130, 9), np.float32), ((5, 33, 100, 5), np.float64)])
def test_fused_4d_path_equals_generic_nd(shape, dt, monkeypatch):
    """D = 4 runs slice by slice on the 3-D tile code (decompose_fused4: even slices of the
    slowest dim as 3-D passes, odd slices interpolating across t as well, then the t-sweep and
    four Thomas solves). MGH_FUSED4=0 keeps the generic one-thread-per-element N-D kernels:
    same coefficients, integers and outliers, bit for bit -- and both equal the oracle."""
    torch, mg = _gpu()
    u = smooth_field(shape, dt, noise=1e-2)
    ud = torch.from_numpy(u).cuda()
    o = oracle.Hierarchy(shape, dt)
    ref = o.decompose(u)
    h = mg.Hierarchy(shape, dt)
    c = h.decompose(ud)
    assert_bit_equal(c.cpu().numpy(), ref, "fused 4-D decompose %r" % (shape,))
    q, oi, ov, cnt, nrm = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf, outlier_cap=u.size)
    # the way back on the same slice-by-slice level loop (recompose_levels4): coefficients ->
    # nodal values equal to the oracle's recomposition, out of place and in place; integers ->
    # reconstruction
    ref_back = o.recompose(ref)
    assert_bit_equal(h.recompose(c).cpu().numpy(), ref_back, "fused 4-D recompose %r" % (shape,))
    c_in = c.clone()
    assert_bit_equal(h.recompose(c_in, out=c_in).cpu().numpy(), ref_back, "fused 4-D recompose in place")
    back = h.dequantize_recompose(q.clone(), mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi[:cnt], outlier_val=ov[:cnt])
    h.close()
    # (round 6: all t-slices of a kind in one launch; MGH_SLICE_BATCH=0: the launch per slice it replaced)
    monkeypatch.setenv("MGH_SLICE_BATCH", "0")
    p = mg.Hierarchy(shape, dt)
    assert_bit_equal(p.recompose(c).cpu().numpy(), ref_back, "fused 4-D recompose, a launch per slice %r" % (shape,))
    assert torch.equal(back, p.dequantize_recompose(q.clone(), mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi[:cnt],
                                                    outlier_val=ov[:cnt]))
    p.close()
    monkeypatch.delenv("MGH_SLICE_BATCH")
    monkeypatch.setenv("MGH_FUSED4", "0")
    g = mg.Hierarchy(shape, dt)
    q2, oi2, ov2, cnt2, nrm2 = g.decompose_quantize(ud, mg.REL, 1e-3, np.inf, outlier_cap=u.size)
    assert cnt == cnt2 and nrm == nrm2 and torch.equal(q, q2)
    a, b = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy()), _outlier_set(oi2.cpu().numpy(), ov2.cpu().numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    back2 = g.dequantize_recompose(q2.clone(), mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi2[:cnt2], outlier_val=ov2[:cnt2])
    assert torch.equal(back, back2)
    assert float((back - ud).abs().max().item()) <= 1e-3 * nrm
    g.close()


def test_config3_full_size_slab():
    """BASELINE.json configs[3] at FULL per-rank size, 8 x 512^3 float32: the fused 4-D path
    against the generic N-D kernels bit for bit, against the ORACLE's N-D path at full size where
    the host has the memory and the cores for it (OpenMP over pencils: seconds on the 128-core
    GPU hosts), and the round trip through the slice-by-slice recomposition within the tolerance."""
    torch, mg = _gpu()
    shape = (8, 512, 512, 512)
    if _host_mem_gb() < 24 or torch.cuda.mem_get_info()[0] < (40 << 30):
        pytest.skip("not enough memory for the full-size slab")
    u = _slab4d(shape)
    ud = torch.from_numpy(u).cuda()
    nrm = float(np.max(np.abs(u)))
    cap = ud.numel() // 8
    h = mg.Hierarchy(shape, np.float32)
    assert h.l_target == 3
    q, oi, ov, cnt, n1 = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf, outlier_cap=cap)
    assert n1 == nrm and cnt <= cap
    if _host_mem_gb() >= 96 and (os.cpu_count() or 1) >= 32:
        o = oracle.Hierarchy(shape, np.float32)
        rq, roi, rov, rn = o.quantize(o.decompose(u), oracle.REL, np.float32(1e-3), np.float32(np.inf),
                                      np.float32(nrm), outlier_cap=cap)
        assert cnt == rn
        assert np.array_equal(q.cpu().numpy(), rq)
        gi, gv = _outlier_set(oi.cpu().numpy()[:cnt], ov.cpu().numpy()[:cnt])
        ri, rv = _outlier_set(roi[:rn], rov[:rn])
        assert np.array_equal(gi, ri) and np.array_equal(gv, rv)
        del o, rq, roi, rov
    del u
    os.environ["MGH_FUSED4"] = "0"
    try:
        g = mg.Hierarchy(shape, np.float32)
    finally:
        del os.environ["MGH_FUSED4"]
    q2, oi2, ov2, cnt2, n2 = g.decompose_quantize(ud, mg.REL, 1e-3, np.inf, outlier_cap=cap)
    assert cnt == cnt2 and torch.equal(q, q2)
    a, b = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy()), _outlier_set(oi2.cpu().numpy(), ov2.cpu().numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    del q2, oi2, ov2
    g.close()
    back = h.dequantize_recompose(q, mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    assert float((back - ud).abs().max().item()) <= 1e-3 * nrm
    h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("env", [
    {"MGH_BOX": "0"}, {"MGH_BOX": "2"}, {"MGH_BOX": "3", "MGH_TAIL_SOLVES": "0"},
    {"MGH_IPK_WPC": "1"}, {"MGH_IPK_WPC": "16", "MGH_CLS1": "0"}])
@pytest.mark.parametrize("shape,mode,s,dict_size", [((130, 97, 258), "REL", np.inf, 8192), ((67, 66, 65), "ABS", 0.0, 64),
                                                    ((36, 260, 100), "REL", np.inf, 64)])
def test_round3_schedules_bit_exact(shape, mode, s, dict_size, env, dt, monkeypatch):
    """The round-3 pieces against the oracle, each switched on every level it can run on and
    switched off: the box kernel (kernels_box.hpp: no march, workgroup-wide outlier slots; the
    small dictionary makes most values outliers), the tail kernel with / without the Thomas
    solves of the level above it, the residency plans of the streaming Thomas solves. Same
    integers, same outlier set."""
    torch, mg = _gpu()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    u = smooth_field(shape, dt, noise=3e-3)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    m = mg.REL if mode == "REL" else mg.ABS
    nrm = float(np.max(np.abs(u))) if mode == "REL" else 1.0
    q, oi, ov, cnt, _ = h.decompose_quantize(ud, m, 1e-3, s, norm=nrm, dict_size=dict_size, outlier_cap=u.size)
    rq, roi, rov, rn = o.quantize(o.decompose(u), oracle.REL if mode == "REL" else oracle.ABS,
                                  dt(1e-3), dt(s), dt(nrm), dict_size=dict_size, outlier_cap=u.size)
    assert cnt == rn
    assert np.array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    assert np.array_equal(gi, ri) and np.array_equal(gv, rv)
    # floating-point coefficients (OUT_T variants of the same kernels)
    c = h.decompose(ud)
    assert_bit_equal(c.cpu().numpy(), o.decompose(u), "coefficients")
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dt,mode,tol,s,dict_size", [
    ((65, 70, 129), np.float32, "REL", 1e-3, np.inf, 8192), ((33, 40, 36), np.float64, "ABS", 1e-4, 0.0, 8192),
    ((129, 129, 129), np.float32, "REL", 1e-4, np.inf, 64), ((20, 17, 300), np.float32, "REL", 1e-2, 1.0, 65536),
    ((8, 66, 70, 129), np.float32, "REL", 1e-3, np.inf, 8192), ((7, 33, 40, 65), np.float64, "ABS", 1e-5, np.inf, 64)])
@pytest.mark.parametrize("mixed", ["1", "0"])
def test_sym16_output_equals_the_int64_output(shape, dt, mode, tol, s, dict_size, mixed, monkeypatch):
    """mgh_decompose_quantize_sym16 = mgh_decompose_quantize(prep_huffman=1) narrowed to 16 bits,
    same outlier list (the small dictionary forces many outliers). The way back with the symbol
    width chosen per level (finest level: 16-bit symbols + outlier table, below: int64 copy of the
    coarse corner box; the default from 2^18 elements on) and with 16-bit symbols on every level
    (MGH_SYM16_MIXED=0; read when the hierarchy is created)."""
    import torch
    import mgard_amd as mg
    monkeypatch.setenv("MGH_SYM16_MIXED", mixed)
    u = smooth_field(shape, dt)
    d = torch.from_numpy(u).cuda()
    h = mg.Hierarchy(shape, dt)
    eb = getattr(mg, mode)
    norm = 0.0 if (mode == "REL" and np.isinf(s)) else (float(np.sqrt(np.mean(u.astype(np.float64) ** 2))) if mode == "REL" else 0.0)
    q, oi, ov, cnt, n1 = h.decompose_quantize(d, eb, tol, s, norm=norm, dict_size=dict_size)
    sym, si, sv, scnt, n2 = h.decompose_quantize_sym16(d, eb, tol, s, norm=norm, dict_size=dict_size)
    assert cnt == scnt and n1 == n2
    assert np.array_equal(sym.cpu().numpy().astype(np.int64), q.cpu().numpy())
    a = sorted(zip(oi.cpu().numpy().tolist(), ov.cpu().numpy().tolist()))
    b = sorted(zip(si.cpu().numpy().tolist(), sv.cpu().numpy().tolist()))
    assert a == b
    if dict_size == 64:
        assert cnt > 100
    # and back: the 16-bit path reconstructs the same bits as the int64 path
    nrm = n1 if mode == "REL" else 0.0
    v64 = h.dequantize_recompose(q.clone(), eb, tol, s, nrm, dict_size=dict_size, outlier_idx=oi, outlier_val=ov)
    v16 = h.dequantize_recompose_sym16(sym, eb, tol, s, nrm, dict_size=dict_size, outlier_idx=si, outlier_val=sv)
    assert np.array_equal(v64.cpu().numpy().view(np.uint8), v16.cpu().numpy().view(np.uint8))
    # out-of-range and duplicate outlier indices (damaged stream) must not fault
    if cnt:
        bad_i = torch.cat([si, torch.tensor([h.total + 5, 2 ** 40], dtype=si.dtype, device=si.device), si[:1]])
        bad_v = torch.cat([sv, torch.tensor([7, 8], dtype=sv.dtype, device=sv.device), sv[:1]])
        h.dequantize_recompose_sym16(sym, eb, tol, s, nrm, dict_size=dict_size, outlier_idx=bad_i, outlier_val=bad_v)
        torch.cuda.synchronize()
    h.close()
    # not on the generic paths
    h2 = mg.Hierarchy((300, 40), np.float32)
    with pytest.raises(mg.MgardHipError):
        h2.decompose_quantize_sym16(torch.zeros((300, 40), device="cuda"), mg.ABS, 1e-3, np.inf)
    h2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_even_size_ghost_node_against_reference_number_gpu(dt):
    """The even-size (ghost node) case tied to the reference's own golden through the even = odd
    embedding (derivation: tests/test_oracle_goldens.py) -- through the HIP path."""
    from tests.test_oracle_goldens import _even4_expected, approx
    torch, mg = _gpu()
    x_even, expected = _even4_expected()
    h = mg.Hierarchy((4,), dt, coords=[np.array([0, 1, 2, 4], dtype=dt)])
    got = h.decompose(torch.from_numpy(x_even.astype(dt)).cuda()).cpu().numpy()
    assert approx(got, expected), (got, expected)
    back = h.recompose(torch.from_numpy(expected.astype(dt)).cuda()).cpu().numpy()
    assert approx(back, x_even)
    h.close()


@pytest.mark.gpu
def test_quantizer_known_answers_gpu():
    """tests/src/test_LinearQuantizer.cpp:94-111 through mgh_quantize / mgh_dequantize."""
    torch, mg = _gpu()
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))["quantizer"]
    qz, dq = G["quantize"], G["dequantize"]
    h = mg.Hierarchy((5,), np.float64)
    x = torch.tensor(qz["x"], dtype=torch.float64, device="cuda")
    q, oi, ov, n = h.quantize(x, mg.ABS, 6 * qz["quantum"], np.inf, 1.0, prep_huffman=False)
    assert n == 0 and q.cpu().tolist() == qz["n"]
    h.close()
    h = mg.Hierarchy((5,), np.float32)
    ns = torch.tensor(dq["n"] + [0], dtype=torch.int64, device="cuda")
    back = h.dequantize(ns, mg.ABS, 6 * dq["quantum"], np.inf, 1.0, prep_huffman=False)
    assert back.cpu().tolist() == dq["x"] + [0.0]
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(9,), (8,), (1000,), (5, 9), (33, 20), (5, 9, 17), (6, 8, 10), (65, 70, 129),
                                   (5, 5, 5, 5), (7, 6, 9, 12), (3, 4, 5, 6, 7)])
def test_level_linearize_equals_the_oracle(shape):
    """config.reorder == 1 (LinearQuantization.hpp:46-146, 588-605): mgh_level_linearize is the
    permutation the restated calc_level_offset defines, its inverse undoes it, and outlier indices
    map to the positions of their values."""
    import torch
    import mgard_amd as mg
    h = mg.Hierarchy(shape, np.float32)
    o = oracle.Hierarchy(shape, np.float32)
    n = int(np.prod(shape))
    q = np.arange(n, dtype=np.int64).reshape(shape) * 3 - 7
    ref = o.level_linearize(q)
    assert np.array_equal(np.sort(ref), np.sort(q.reshape(-1)))          # a permutation
    qd = torch.from_numpy(q).cuda()
    rng = np.random.default_rng(4)
    oi = rng.choice(n, size=min(n, 50), replace=False).astype(np.int64)
    oid = torch.from_numpy(oi.copy()).cuda()
    lin = h.level_linearize(qd, outlier_idx=oid)
    assert np.array_equal(lin.cpu().numpy().reshape(-1), ref)
    assert np.array_equal(oid.cpu().numpy(), np.array([o.linearized_position(i) for i in oi], dtype=np.int64))
    assert np.array_equal(lin.cpu().numpy().reshape(-1)[oid.cpu().numpy()], q.reshape(-1)[oi])
    back = h.level_linearize(lin, inverse=True)
    assert torch.equal(back, qd)
    # level 0 comes first and keeps its order: the first entries are the coarsest nodes
    l0 = int(np.prod(o.level_shape(0)))
    h.close()
    assert l0 >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dt,nonuni,mode,s", [
    ((258, 515, 517), np.float32, False, "REL", np.inf),
    ((260, 515, 520), np.float64, True, "ABS", 0.0),
    ((257, 516, 513), np.float32, True, "ABS", np.inf)])
def test_long_march_class_with_odd_remainders(shape, dt, nonuni, mode, s):
    """Shapes big enough for the long-march class of the level kernel (4 x 64 tiles, chunks of 16
    coarse planes, XCD ranges) whose coarse sizes are NOT tile multiples plus one: partial main
    tiles, f- and c-face tiles with 1-4 columns / rows, an even and an odd slowest dimension (last
    chunk with and without the extra plane), ghost nodes in some dims. Bit-exact integers and
    outlier sets against the oracle, reconstruction through the fused way back."""
    torch, mg = _gpu()
    if _host_mem_gb() < 24:
        pytest.skip("not enough host memory for the oracle run")
    u = smooth_field(shape, dt)
    coords = nonuniform_coords(shape, dt) if nonuni else None
    h = mg.Hierarchy(shape, dt, coords=coords)
    o = oracle.Hierarchy(shape, dt, coords=coords)
    ud = torch.from_numpy(u).cuda()
    eb = mg.REL if mode == "REL" else mg.ABS
    nrm = float(np.max(np.abs(u))) if mode == "REL" else 1.0
    q, oi, ov = _compare_quantized(mg, h, o, u, ud, eb, 1e-3, s, nrm, u.size // 8)
    back = h.dequantize_recompose(q.clone(), eb, 1e-3, s, nrm, outlier_idx=oi, outlier_val=ov)
    if np.isinf(s):
        assert float((back - ud).abs().max().item()) <= 1e-3 * nrm
    c = h.decompose(ud)
    assert_bit_equal(h.recompose(c).cpu().numpy(), o.recompose(o.decompose(u)), "recompose %r" % (shape,))
    h.close()


@pytest.mark.gpu
def test_4d_long_march_class_with_odd_remainders():
    """D = 4 with slices big enough for the long-march class (4 x 64 tiles) and sizes that leave
    partial / face tiles and ghost nodes in every dimension, odd number of t-slices: the
    slice-by-slice path against the generic N-D kernels, both directions, bit for bit."""
    torch, mg = _gpu()
    shape = (5, 258, 515, 517)
    if torch.cuda.mem_get_info()[0] < (24 << 30):
        pytest.skip("not enough device memory")
    u = _slab4d(shape)
    ud = torch.from_numpy(u).cuda()
    nrm = float(np.max(np.abs(u)))
    del u
    cap = ud.numel() // 4
    h = mg.Hierarchy(shape, np.float32)
    q, oi, ov, cnt, n1 = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf, outlier_cap=cap)
    back = h.dequantize_recompose(q.clone(), mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    os.environ["MGH_FUSED4"] = "0"
    try:
        g = mg.Hierarchy(shape, np.float32)
    finally:
        del os.environ["MGH_FUSED4"]
    q2, oi2, ov2, cnt2, n2 = g.decompose_quantize(ud, mg.REL, 1e-3, np.inf, outlier_cap=cap)
    assert cnt == cnt2 and cnt <= cap and n1 == n2 == nrm and torch.equal(q, q2)
    a, b = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy()), _outlier_set(oi2.cpu().numpy(), ov2.cpu().numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    back2 = g.dequantize_recompose(q2, mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi2, outlier_val=ov2)
    assert torch.equal(back, back2)
    assert float((back - ud).abs().max().item()) <= 1e-3 * nrm
    g.close()
    h.close()


@pytest.mark.parametrize("dt", [np.float32])
@pytest.mark.parametrize("shape", [(40, 330, 70), (400, 30, 50), (330, 200, 36), (19, 321, 130), (513, 9, 70),
                                   (12, 320, 9, 33)])
def test_ipk_dma_on_small_shapes(shape, dt, monkeypatch):
    """k_ipk_dma (LDS-DMA front end of the strided Thomas solves) normally runs only on levels of
    512+ tiles; MGH_IPK_DMA_MIN=0 puts every float level whose pencils are long enough (160+
    elements) on it: LDS parts of 1 ... 97 rows (walked singly / in batches / none), the four-rows-
    per-instruction DMA with its overlapping tail, tiles that straddle a plane or end the array
    (one row per instruction), a last tile of a single pencil, AddND (+) and SubtractND (-) fused.
    Decomposition and recomposition bit-identical to the oracle."""
    torch, mg = _gpu()
    monkeypatch.setenv("MGH_IPK_DMA_MIN", "0")
    u = smooth_field(shape, dt, noise=3e-3)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ref = o.decompose(u)
    c = h.decompose(torch.from_numpy(u).cuda())
    assert_bit_equal(c.cpu().numpy(), ref, "decompose")
    back = h.recompose(c)
    assert_bit_equal(back.cpu().numpy(), o.recompose(ref), "recompose")
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dt", [((130, 129, 257), np.float32), ((66, 131, 260), np.float64),
                                      ((8, 40, 66, 129), np.float32)])
def test_outputs_at_odd_offsets_equal_the_aligned_ones(shape, dt):
    """The level pass lays its stores out for a 128-byte aligned output array (a lane stores the
    odd-f coefficients of the cell to its left so that a wave's row pieces start on line
    boundaries); any element-aligned base must give the same integers, coefficients and outlier
    sets: int64 output at +8 / +504 bytes, float coefficients at +4 / +12 elements, the input at
    an odd element offset too, and the way back reads the shifted arrays."""
    torch, mg = _gpu()
    u = smooth_field(shape, dt)
    N = u.size
    tdt = torch.float32 if dt == np.float32 else torch.float64
    ud = torch.from_numpy(u).cuda()
    h = mg.Hierarchy(shape, dt)
    nrm = float(np.max(np.abs(u)))
    cap = N  # (a list that overflows holds an arbitrary subset)
    q0, oi0, ov0, n0, _ = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf, nrm, outlier_cap=cap)
    assert n0 <= cap
    c0 = h.decompose(ud)
    order0 = torch.argsort(oi0)
    for q_off, c_off, u_off in ((1, 1, 3), (63, 3, 1)):
        upool = torch.empty(N + 64, dtype=tdt, device="cuda")
        uv = upool[u_off:u_off + N].view(shape)
        uv.copy_(ud)
        assert h.norm(uv) == h.norm(ud) == nrm
        qpool = torch.full((N + 128,), -7, dtype=torch.int64, device="cuda")
        qv = qpool[q_off:q_off + N].view(shape)
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        oi = torch.empty(cap, dtype=torch.int64, device="cuda")
        ov = torch.empty(cap, dtype=torch.int64, device="cuda")
        h.decompose_quantize(uv, mg.REL, 1e-3, np.inf, nrm, bufs=(qv, cnt, oi, ov), want_norm=False)
        torch.cuda.synchronize()
        assert int(cnt.item()) == n0
        assert torch.equal(qv, q0), "int64 output at +%d bytes differs" % (8 * q_off)
        # nothing outside the view was touched
        assert int((qpool[:q_off] != -7).sum().item()) == 0 and int((qpool[q_off + N:] != -7).sum().item()) == 0
        order = torch.argsort(oi[:n0])
        assert torch.equal(oi[:n0][order], oi0[order0]) and torch.equal(ov[:n0][order], ov0[order0])
        cpool = torch.zeros(N + 64, dtype=tdt, device="cuda")
        cv = cpool[c_off:c_off + N].view(shape)
        h.decompose(uv, out=cv)
        assert_bit_equal(cv.cpu().numpy(), c0.cpu().numpy(), "coefficients at +%d elements" % c_off)
        back = h.dequantize_recompose(qv, mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi[:n0], outlier_val=ov[:n0])
        back0 = h.dequantize_recompose(q0.clone(), mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi0, outlier_val=ov0)
        assert_bit_equal(back.cpu().numpy(), back0.cpu().numpy(), "reconstruction from the shifted integers")
    h.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_decompose_equals_the_operator_built_decomposition_on_nonuniform_grids(dt):
    """The -m gpu mirror of tests/test_oracle_operator_goldens.py step 3: mgh_decompose against a
    decomposition assembled from the dense mass / interpolation / restriction matrices that the
    reference's own operator vectors pin (tests/src/test_TensorMassMatrix.cpp:21-262,
    test_TensorRestriction.cpp:18-221, test_TensorProlongation.cpp:16-106), on non-uniform dyadic
    grids in 1-3 D -- the reference's 5 x 5 custom spacing among them. Tolerance: rounding of the
    data type (the matrices are evaluated in double)."""
    from tests.test_oracle_operator_goldens import NONUNIFORM_GRIDS, decompose_by_matrices, nonuniform_grid
    torch, mg = _gpu()
    tol = 5e-5 if dt == np.float32 else 1e-11
    for shape, seed in NONUNIFORM_GRIDS:
        coords = [x.astype(dt) for x in nonuniform_grid(shape, seed)]
        u = np.random.default_rng(11).normal(size=shape).astype(dt)
        h = mg.Hierarchy(shape, dt, coords=coords)
        got = oracle.dyadic_reordered_to_natural(h.decompose(torch.from_numpy(u).cuda()).cpu().numpy())
        want = decompose_by_matrices([c.astype(np.float64) for c in coords], u.astype(np.float64))
        assert np.allclose(got.astype(np.float64), want, rtol=tol, atol=tol * np.abs(want).max()), shape
        # (and bit for bit what the oracle computes: the product and its checker agree on the very
        # grids the reference's numbers were checked on)
        o = oracle.Hierarchy(shape, dt, coords=coords)
        assert_bit_equal(h.decompose(torch.from_numpy(u).cuda()).cpu().numpy(), o.decompose(u), "decompose %r" % (shape,))
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"MGH_IPK_SPEC_K": "2"}, {"MGH_IPK_SPEC_K": "9"}, {"MGH_IPK_SPEC": "0"},
                                 {"MGH_IPK_SPEC_LONG": "0"}])
@pytest.mark.parametrize("shape,dt,nonuniform", [((1 << 20,), np.float32, False), ((300001,), np.float64, True),
                                                 ((3, 40000), np.float32, False), ((70001,), np.float32, True),
                                                 ((40000, 3), np.float32, False), ((20001, 4, 5), np.float64, True),
                                                 ((6, 9000, 7), np.float32, False), ((70, 9, 5000), np.float32, True),
                                                 # (round 6: strided pencils of 1024 and more that WOULD fit LDS, one
                                                 # round of tiles: chunked too -- MGH_IPK_SPEC_LONG)
                                                 ((2500, 6, 7), np.float32, False), ((5, 2300, 9), np.float64, True),
                                                 ((2100, 33, 40), np.float32, True)])
def test_long_contiguous_pencils_are_solved_in_verified_chunks(shape, dt, nonuniform, env, monkeypatch):
    """Few long pencils (a 1-D array is one pencil per level; 40000 x 3 has three strided ones; pencils too
    long for LDS in general) are solved in chunks that start
    from a wrong state a warm-up length in front of their first element; every chunk's start is then
    compared bit for bit with the end of the chunk before it and recomputed where they differ
    (kernels_ipk_spec.hpp). The result must be the sequential sweep's (IPKFunctor.h:111-149) whatever the
    warm-up length: default, 2 and 9 (nearly every chunk fails the comparison and is repaired), and with
    the one-lane-per-pencil kernel (MGH_IPK_SPEC=0)."""
    torch, mg = _gpu()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    coords = nonuniform_coords(shape, dt) if nonuniform else None
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt, coords=coords)
    o = oracle.Hierarchy(shape, dt, coords=coords)
    ud = torch.from_numpy(u).cuda()
    c = h.decompose(ud)
    ref = o.decompose(u)
    assert_bit_equal(c.cpu().numpy(), ref, "decompose %r %r" % (shape, env))
    assert_bit_equal(h.recompose(c).cpu().numpy(), o.recompose(ref), "recompose %r %r" % (shape, env))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"MGH_IPK_CHUNK": "1"}, {"MGH_IPK_CHUNK": "1", "MGH_IPK_CHUNK_K": "3"},
                                 {"MGH_IPK_CHUNK": "1", "MGH_IPK_CHUNK_K": "12"}, {"MGH_IPK_CHUNK": "0"},
                                 {"MGH_IPK_CHUNK": "1", "MGH_FORCE_V1": "1"},
                                 {"MGH_IPK_CHUNK": "1", "MGH_IPK_CHUNK_K": "5", "MGH_FORCE_V1": "1"}])
@pytest.mark.parametrize("shape,dt,nonuniform", [((129, 129, 257), np.float32, False), ((130, 67, 200), np.float64, True),
                                                 ((257, 140, 131), np.float32, True), ((96, 300, 128), np.float64, False),
                                                 ((5, 70, 66, 129), np.float32, False)])
def test_tiles_in_lds_are_solved_in_verified_chunks(shape, dt, nonuniform, env, monkeypatch):
    """The LDS-staged Thomas solve of contiguous pencils shares the two sweeps of a tile between the four
    waves of the workgroup (kernels_ipk.hpp:thomas_chunked): a lane sweeps one quarter of a pencil from a wrong state a warm-up
    length in front of it, every quarter's start is compared bit for bit with the end of the quarter
    before it, and a tile with a mismatch is loaded again and solved by one lane per pencil. The result
    must be the sequential sweep's (IPKFunctor.h:111-149) whatever the warm-up length: the one derived
    from the tables, 3, 5 and 12 (most tiles fail the comparison and take the second pass), with the
    kernels that do not chunk, and on the unfused path (MGH_FORCE_V1: one solve per axis at every level,
    where the fused path solves f and c of the small levels in one plane kernel)."""
    torch, mg = _gpu()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    coords = nonuniform_coords(shape, dt) if nonuniform else None
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt, coords=coords)
    o = oracle.Hierarchy(shape, dt, coords=coords)
    ud = torch.from_numpy(u).cuda()
    c = h.decompose(ud)
    ref = o.decompose(u)
    assert_bit_equal(c.cpu().numpy(), ref, "decompose %r %r" % (shape, env))
    assert_bit_equal(h.recompose(c).cpu().numpy(), o.recompose(ref), "recompose %r %r" % (shape, env))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("agg", ["0", "1", "2"])
@pytest.mark.parametrize("tol", [1e-3, 1e-5, 1e-7])
@pytest.mark.parametrize("shape,dt", [((65, 70, 129), np.float32), ((130, 67, 200), np.float64),
                                      ((129, 129, 257), np.float32), ((3, 40, 66, 65), np.float32),
                                      ((6, 33, 40, 70), np.float64), ((33, 34),  np.float32)])
def test_outlier_slots_per_wave_and_per_workgroup_fill_the_same_list(shape, dt, tol, agg, monkeypatch):
    """The level kernel has two ways of asking for slots in the outlier list: per wave and plane, and
    (kernels_fused2.hpp: OutlierShared) once per workgroup and pair of planes from a stash in LDS, for
    fields where most values leave the dictionary. MGH_OUTLIER_AGG = 0 / 1 fixes the variant, 2 (the
    default) lets the previous call's outlier count decide: the second and third call below run the
    other variant when the tolerance is tight. Quantized values and the outlier SET must be the
    oracle's in every case (LinearQuantization.hpp:208-241; the order of the list is not defined)."""
    torch, mg = _gpu()
    monkeypatch.setenv("MGH_OUTLIER_AGG", agg)
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    nrm = oracle.norm(u, dt(np.inf))
    rq, roi, rov, rn = o.quantize(o.decompose(u), oracle.REL, dt(tol), dt(np.inf), dt(nrm))
    ri, rv = _outlier_set(roi, rov)
    for call in range(3):
        q, oi, ov, n, got_nrm = h.decompose_quantize(ud, mg.REL, tol, np.inf)
        torch.cuda.synchronize()
        assert got_nrm == nrm and n == rn, (call, n, rn)
        np.testing.assert_array_equal(q.cpu().numpy(), rq)
        gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
        np.testing.assert_array_equal(gi, ri)
        np.testing.assert_array_equal(gv, rv)
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("v1", [None, "0", "1"])
@pytest.mark.parametrize("shape,dt", [((5000, 5, 7), np.float32), ((3000, 17, 17), np.float64), ((40001, 3, 3), np.float32)])
def test_thin_arrays_take_the_simple_kernels_and_give_the_same_bits(shape, dt, v1, monkeypatch):
    """A long slowest dimension over planes of a few nodes fills the tiles of the level kernels badly; the
    hierarchy then selects the one-thread-per-element kernels by itself (mgh_hierarchy_create;
    MGH_FORCE_V1=0 keeps the tiled kernels, =1 forces the simple ones everywhere). Whatever runs, the
    integers are the oracle's."""
    torch, mg = _gpu()
    if v1 is None:
        monkeypatch.delenv("MGH_FORCE_V1", raising=False)
    else:
        monkeypatch.setenv("MGH_FORCE_V1", v1)
    u = smooth_field(shape, dt, noise=1e-2)
    h = mg.Hierarchy(shape, dt)
    o = oracle.Hierarchy(shape, dt)
    ud = torch.from_numpy(u).cuda()
    ref = o.decompose(u)
    assert_bit_equal(h.decompose(ud).cpu().numpy(), ref, "decompose %r" % (shape,))
    q, oi, ov, n, nrm = h.decompose_quantize(ud, mg.REL, 1e-3, np.inf)
    rq, roi, rov, rn = o.quantize(ref, oracle.REL, dt(1e-3), dt(np.inf), dt(nrm))
    assert n == rn
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    gi, gv = _outlier_set(oi.cpu().numpy(), ov.cpu().numpy())
    ri, rv = _outlier_set(roi, rov)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gv, rv)
    back = h.dequantize_recompose(q, mg.REL, 1e-3, np.inf, nrm, outlier_idx=oi, outlier_val=ov)
    rback = o.recompose(o.dequantize(rq, oracle.REL, dt(1e-3), dt(np.inf), dt(nrm), outlier_idx=roi, outlier_val=rov))
    assert_bit_equal(back.cpu().numpy(), rback, "dequantize + recompose %r" % (shape,))
    h.close()
