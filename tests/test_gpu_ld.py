"""GPU tests of mgh_set_ld: the low-level C ABI on PITCHED arrays of the data type (the reference's
HIP backend allocates its Arrays with hipMallocPitch by default: RuntimeX/DataStructures/Array.hpp:
70-84, SubArray.hpp:136-139; DataRefactor.hpp:19-152 and LinearQuantization.hpp run on such arrays).
Every call on a pitched array must give what the same call gives on the dense array, and must not
touch the padding."""
import numpy as np
import pytest

from tests.util import smooth_field

pytestmark = pytest.mark.gpu

SHAPES = [
    ((33, 40, 65), "fused 3-D"),
    ((129, 66, 130), "fused 3-D, several levels"),
    ((40, 130, 9), "fused 3-D, 64 x 4 tiles (short fastest extent)"),
    ((300, 5, 7), "thin 3-D: one-thread-per-element kernels"),
    ((100, 129), "2-D"),
    ((5, 9, 10, 17), "fused 4-D"),
    ((3, 4, 5, 6, 7), "5-D generic"),
]


def _mods():
    import torch
    import mgard_amd
    return torch, mgard_amd


def _ld(shape, itemsize, pad_mid):
    """Row pitch rounded up to 256 bytes (what hipMallocPitch does); pad_mid: dimension D-2 padded by
    three rows as well (SubArray carries one ld per dimension)."""
    ld = list(shape)
    per = 256 // itemsize
    ld[-1] = (shape[-1] + per - 1) // per * per
    if ld[-1] == shape[-1]:
        ld[-1] += per
    if pad_mid and len(shape) >= 2:
        ld[-2] = shape[-2] + 3
    return ld


def _pitched(torch, u, ld, fill=float("nan")):
    shape = u.shape
    full = torch.full([shape[0]] + list(ld[1:]), fill, dtype=u.dtype, device=u.device)
    full[tuple(slice(0, n) for n in shape)] = u
    return full


def _valid(t, shape):
    return t[tuple(slice(0, n) for n in shape)]


def _padding_untouched(torch, t, shape):
    mask = torch.ones_like(t, dtype=torch.bool)
    mask[tuple(slice(0, n) for n in shape)] = False
    return bool(torch.isnan(t[mask]).all())


@pytest.mark.parametrize("pad_mid", [False, True])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [s for s, _ in SHAPES], ids=[n for _, n in SHAPES])
def test_every_stage_on_pitched_arrays_equals_the_dense_result(shape, dt, pad_mid):
    torch, mg = _mods()
    u = torch.from_numpy(smooth_field(shape, dt, noise=1e-3)).cuda()
    ld = _ld(shape, np.dtype(dt).itemsize, pad_mid)
    h = mg.Hierarchy(shape, dt)
    # dense reference results
    nrm = h.norm(u, float("inf"))
    nrm2 = h.norm(u, 0.0)
    coef = h.decompose(u)
    q, oi, ov, cnt = h.quantize(coef, mg.REL, 1e-3, float("inf"), nrm, dict_size=64)
    fq, foi, fov, fcnt, fn = h.decompose_quantize(u, mg.REL, 1e-3, float("inf"), dict_size=64)
    back = h.dequantize_recompose(fq.clone(), mg.REL, 1e-3, float("inf"), fn, dict_size=64, outlier_idx=foi,
                                  outlier_val=fov)
    deq = h.dequantize(q.clone(), mg.REL, 1e-3, float("inf"), nrm, dict_size=64, outlier_idx=oi, outlier_val=ov)
    rec = h.recompose(coef)

    # ---- pitched input ----
    up = _pitched(torch, u, ld)
    cp = _pitched(torch, coef, ld)
    h.set_ld(mg.LD_IN, ld)
    assert h.norm(up, float("inf")) == nrm
    # (a sum in another order: rows instead of 16-byte pieces, atomics in whatever order they arrive)
    assert abs(h.norm(up, 0.0) - nrm2) <= 256 * np.finfo(dt).eps * nrm2
    out = torch.empty_like(u)
    h.decompose(up, out=out)
    assert torch.equal(out, coef)
    q2, oi2, ov2, cnt2 = h.quantize(cp, mg.REL, 1e-3, float("inf"), nrm, dict_size=64)
    assert torch.equal(q2, q) and cnt2 == cnt
    assert sorted(zip(oi2.tolist(), ov2.tolist())) == sorted(zip(oi.tolist(), ov.tolist()))
    g = h.decompose_quantize(up, mg.REL, 1e-3, float("inf"), dict_size=64)
    assert torch.equal(g[0], fq) and g[3] == fcnt and g[4] == fn
    assert sorted(zip(g[1].tolist(), g[2].tolist())) == sorted(zip(foi.tolist(), fov.tolist()))
    g = h.decompose_quantize(up, mg.ABS, 1e-3 * fn, float("inf"), dict_size=64)   # (no norm pass)
    assert g[3] == fcnt
    if h.sym16_supported():
        h.set_ld(mg.LD_IN, None)
        a = h.decompose_quantize_sym16(u, mg.REL, 1e-3, float("inf"), dict_size=64)
        h.set_ld(mg.LD_IN, ld)
        b = h.decompose_quantize_sym16(up, mg.REL, 1e-3, float("inf"), dict_size=64)
        assert torch.equal(a[0], b[0]) and a[3] == b[3] and a[4] == b[4]
    out = torch.empty_like(u)
    h.recompose(cp, out=out)
    assert torch.equal(out, rec)
    assert _padding_untouched(torch, up, shape) and _padding_untouched(torch, cp, shape)   # inputs are inputs
    h.set_ld(mg.LD_IN, None)

    # ---- pitched output ----
    h.set_ld(mg.LD_OUT, ld)
    for fn_, args, want in (
            (h.decompose, (u,), coef),
            (h.recompose, (coef,), rec)):
        op = torch.full_like(up, float("nan"))
        fn_(*args, out=op)
        assert torch.equal(_valid(op, shape), want) and _padding_untouched(torch, op, shape)
    op = torch.full_like(up, float("nan"))
    h.dequantize(q.clone(), mg.REL, 1e-3, float("inf"), nrm, dict_size=64, outlier_idx=oi, outlier_val=ov, out=op)
    assert torch.equal(_valid(op, shape), deq) and _padding_untouched(torch, op, shape)
    op = torch.full_like(up, float("nan"))
    h.dequantize_recompose(fq.clone(), mg.REL, 1e-3, float("inf"), fn, dict_size=64, outlier_idx=foi, outlier_val=fov,
                           out=op)
    assert torch.equal(_valid(op, shape), back) and _padding_untouched(torch, op, shape)
    if h.sym16_supported():
        h.set_ld(mg.LD_OUT, None)
        sym = h.decompose_quantize_sym16(u, mg.REL, 1e-3, float("inf"), dict_size=64)
        want = h.dequantize_recompose_sym16(sym[0], mg.REL, 1e-3, float("inf"), sym[4], dict_size=64,
                                            outlier_idx=sym[1], outlier_val=sym[2])
        h.set_ld(mg.LD_OUT, ld)
        op = torch.full_like(up, float("nan"))
        h.dequantize_recompose_sym16(sym[0], mg.REL, 1e-3, float("inf"), sym[4], dict_size=64, outlier_idx=sym[1],
                                     outlier_val=sym[2], out=op)
        assert torch.equal(_valid(op, shape), want) and _padding_untouched(torch, op, shape)

    # ---- both at once, in place on one pitched array (DataRefactor::Decompose works in place) ----
    h.set_ld(mg.LD_IN, ld)
    w = up.clone()
    h.decompose(w, out=w)
    assert torch.equal(_valid(w, shape), coef) and _padding_untouched(torch, w, shape)
    h.recompose(w, out=w)
    assert torch.equal(_valid(w, shape), rec) and _padding_untouched(torch, w, shape)
    # ---- and dense again ----
    h.set_ld(mg.LD_IN, None)
    h.set_ld(mg.LD_OUT, None)
    assert torch.equal(h.decompose(u), coef)
    h.close()


def test_set_ld_arguments():
    torch, mg = _mods()
    h = mg.Hierarchy((20, 30, 40), np.float32)
    with pytest.raises(mg.MgardHipError):
        h.set_ld(mg.LD_IN, [20, 30, 39])          # smaller than the extent
    with pytest.raises(mg.MgardHipError):
        h.set_ld(5, [20, 30, 40])
    h.set_ld(mg.LD_IN, [1, 30, 40])               # dense (ld[0] is not used): no effect
    u = torch.from_numpy(smooth_field((20, 30, 40), np.float32)).cuda()
    a = h.norm(u)
    h.set_ld(mg.LD_IN, [20, 30, 64])
    with pytest.raises(mg.MgardHipError):          # parts of a pitched array: refused
        h.norm_stream(torch.zeros(20 * 30 * 64, device="cuda"), float("inf"), [20 * 30 * 64])
    h.set_ld(mg.LD_IN, None)
    assert h.norm(u) == a
    h.close()


def test_pitched_input_at_benchmark_size():
    """512^3 f32 with the row pitch hipMallocPitch would choose for 513-element rows... here 512 + 64:
    the fused top-level pass reads the pitched array in place; symbols equal the dense run's."""
    torch, mg = _mods()
    shape = (512, 512, 512)
    u = torch.from_numpy(smooth_field(shape, np.float32)).cuda()
    h = mg.Hierarchy(shape, np.float32)
    a = h.decompose_quantize_sym16(u, mg.REL, 1e-3, float("inf"))
    ld = [512, 512, 576]
    up = _pitched(torch, u, ld, fill=1e30)   # (a reader that strays into the padding would blow the norm)
    h.set_ld(mg.LD_IN, ld)
    b = h.decompose_quantize_sym16(up, mg.REL, 1e-3, float("inf"))
    assert a[4] == b[4] and a[3] == b[3] and torch.equal(a[0], b[0])
    h.close()
