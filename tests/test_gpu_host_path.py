"""GPU tests of the host-buffer path of mgh_compress / mgh_decompress (round 6): pageable memory
through the pinned ring and the copy pool, caller-registered memory, auto_pin_host_buffers, output
allocated by the library, the norm reduced piece by piece while the input arrives
(mgh_norm_stream_*), two host threads on one device. Reference behaviour:
include/mgard-x/CompressionHighLevel/CompressionHighLevel.hpp:147-191, 281-286, 464-500;
GPUPipelines.hpp:69-207."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle
from tests.util import smooth_field

pytestmark = pytest.mark.gpu


def _mods():
    import torch
    import mgard_amd
    from mgard_amd import highlevel as hl
    return torch, mgard_amd, hl


def _same_container(hl, a, b):
    """Byte for byte, the outlier lists of a record compared sorted (they are in atomic order)."""
    from tests.test_gpu_highlevel import _canonical_records
    ca, cb = _canonical_records(hl, np.asarray(a)), _canonical_records(hl, np.asarray(b))
    return ca[0] == cb[0] and len(ca[1]) == len(cb[1]) and all(x == y for x, y in zip(ca[1], cb[1]))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("parts", [[1.0], [0.5, 0.5], [0.1, 0.2, 0.3, 0.4], [0.999, 0.001]])
@pytest.mark.parametrize("s", [np.inf, 0.0])
def test_streamed_norm_equals_the_one_pass_norm(parts, dt, s):
    """mgh_norm_stream_begin/_add + a fused call with norm = 0 against the same call on its own:
    max|x| is exact whatever the partition, so symbols and outliers are identical; the L2 sum may
    differ in its last bits (like two runs of the one-pass reduction), so only the norm is compared."""
    torch, mg, hl = _mods()
    shape = (40, 33, 65)
    u = smooth_field(shape, dt)
    d = torch.from_numpy(u).cuda()
    h = mg.Hierarchy(shape, dt)
    ref = h.decompose_quantize_sym16(d, mg.REL, 1e-3, float(s))
    n = u.size
    counts = [int(n * f) for f in parts]
    counts[-1] = n - sum(counts[:-1])
    h.norm_stream(d, float(s), counts)
    got = h.decompose_quantize_sym16(d, mg.REL, 1e-3, float(s))
    if s == np.inf:
        assert got[4] == ref[4] and dt(got[4]) == oracle.norm(u, dt(np.inf))
        assert torch.equal(got[0], ref[0]) and got[3] == ref[3]
    else:
        assert abs(got[4] - ref[4]) <= 256 * np.finfo(dt).eps * ref[4]
    # the accumulated value is consumed by that one call: the next call reduces by itself again
    # (a stale slot would double the L2 sum)
    again = h.decompose_quantize_sym16(d, mg.REL, 1e-3, float(s))
    assert abs(again[4] - ref[4]) <= 256 * np.finfo(dt).eps * ref[4]
    # the int64 fused entry takes the streamed norm too
    h.norm_stream(d, float(s), counts)
    q, oi, ov, cnt, nrm = h.decompose_quantize(d, mg.REL, 1e-3, float(s))
    assert abs(nrm - ref[4]) <= 256 * np.finfo(dt).eps * ref[4]
    if s == np.inf:
        assert nrm == ref[4] and cnt == ref[3] and torch.equal(q, ref[0].to(torch.int64))
    h.close()


def test_streamed_norm_is_refused_off_the_fused_path():
    torch, mg, hl = _mods()
    h = mg.Hierarchy((300,), np.float32)
    d = torch.zeros(300, device="cuda")
    with pytest.raises(mg.MgardHipError):
        h.norm_stream(d, float("inf"), [300])
    h.close()


@pytest.mark.parametrize("shape,dt", [((129, 130, 257), np.float32), ((65, 200, 300), np.float64),
                                      ((3000, 2049), np.float32), ((9, 40, 50, 60), np.float32)])
def test_host_stream_equals_device_stream(shape, dt):
    """Pageable host input (ring + streamed norm) and device-resident input write the same container;
    so does registered host memory and auto_pin_host_buffers = 1. Large enough for several ring
    chunks (> 16 MB) in the first two shapes."""
    torch, mg, hl = _mods()
    u = smooth_field(shape, dt)
    dev = hl.compress(torch.from_numpy(u).cuda(), 1e-3, np.inf, mg.REL).cpu().numpy()
    host = hl.compress(u, 1e-3, np.inf, mg.REL)
    assert _same_container(hl, host, dev)
    auto = hl.compress(u, 1e-3, np.inf, mg.REL, config=hl.Config(auto_pin_host_buffers=1))
    assert _same_container(hl, auto, dev)
    up = u.copy()
    hl.pin(up)
    try:
        assert hl.is_pinned(up)
        pinned = hl.compress(up, 1e-3, np.inf, mg.REL)
        assert _same_container(hl, pinned, dev)
    finally:
        hl.unpin(up)
    # and back: pageable out, registered out, auto-pin
    ref = hl.decompress(torch.from_numpy(dev).cuda()).cpu().numpy()
    v = hl.decompress(host)
    assert np.array_equal(v, ref)
    out = np.zeros(shape, dtype=dt)
    hl.pin(out)
    try:
        assert hl.decompress(host, out=out) is out
        assert np.array_equal(out, ref)
    finally:
        hl.unpin(out)
    v2 = hl.decompress(host, config=hl.Config(auto_pin_host_buffers=1))
    assert np.array_equal(v2, ref)
    nrm = float(np.max(np.abs(u)))
    assert float(np.max(np.abs(v - u))) <= 1e-3 * nrm


def test_streamed_norm_switch_gives_the_same_container(monkeypatch):
    torch, mg, hl = _mods()
    u = smooth_field((200, 160, 300), np.float32)   # 38 MB: three ring chunks
    a = hl.compress(u, 1e-3, np.inf, mg.REL)
    monkeypatch.setenv("MGH_HL_STREAM_NORM", "0")
    b = hl.compress(u, 1e-3, np.inf, mg.REL)
    assert _same_container(hl, a, b)
    monkeypatch.delenv("MGH_HL_STREAM_NORM")
    # L2 norm: the streamed sum may differ in its last bits, the error bound holds either way
    c = hl.compress(u, 1e-3, 0.0, mg.REL)
    v = hl.decompress(c)
    l2 = float(np.sqrt(np.mean((v.astype(np.float64) - u) ** 2)))
    assert l2 <= 1e-3 * float(np.sqrt(np.mean(u.astype(np.float64) ** 2)))


def test_library_allocated_host_buffers():
    """output_pre_allocated = 0 with host buffers: the library allocates (huge-page advised, released
    with free()) -- compress and decompress."""
    torch, mg, hl = _mods()
    L = hl._hl()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    shape = (160, 320, 400)   # 82 MB: above the pre-touch threshold
    u = smooth_field(shape, np.float32)
    cfg = hl.Config()
    shp = (C.c_uint64 * 3)(*shape)
    optr, size = C.c_void_p(), C.c_size_t(0)
    mg._check(L.mgh_compress(3, mg.FLOAT, shp, 1e-3, float("inf"), mg.REL, C.c_void_p(u.ctypes.data),
                             C.byref(optr), C.byref(size), None, C.byref(cfg), 0))
    stream = np.ctypeslib.as_array(C.cast(optr, C.POINTER(C.c_uint8)), shape=(size.value,)).copy()
    libc.free(optr)
    assert _same_container(hl, stream, hl.compress(u, 1e-3, np.inf, mg.REL))
    for _ in range(2):
        dptr = C.c_void_p()
        mg._check(L.mgh_decompress(C.c_void_p(stream.ctypes.data), stream.size, C.byref(dptr), C.byref(cfg), 0))
        v = np.ctypeslib.as_array(C.cast(dptr, C.POINTER(C.c_float)), shape=shape).copy()
        libc.free(dptr)
        assert float(np.max(np.abs(v - u))) <= 1e-3 * float(np.max(np.abs(u)))
    # a damaged stream with a library-allocated output: error, nothing leaked to the caller
    bad = stream.copy()
    bad[len(bad) // 2:] = 0
    dptr = C.c_void_p()
    rc = L.mgh_decompress(C.c_void_p(bad.ctypes.data), bad.size - 1000, C.byref(dptr), C.byref(cfg), 0)
    assert rc < 0 and not dptr.value


def test_two_host_threads_on_one_device():
    """Per-thread caches (CompressorCache is thread_local in the reference): two host threads call
    mgh_compress / mgh_decompress on the same device at once, host and device buffers."""
    torch, mg, hl = _mods()
    shapes = [(70, 90, 110), (64, 129, 65)]
    fields = [smooth_field(s, np.float32, seed=7 + i) for i, s in enumerate(shapes)]
    expect = [hl.compress(f, 1e-3, np.inf, mg.REL) for f in fields]
    errors = []

    def work(i):
        try:
            torch.cuda.set_device(0)
            for it in range(6):
                if it % 2 == 0:
                    c = hl.compress(fields[i], 1e-3, np.inf, mg.REL)
                    v = hl.decompress(c)
                else:
                    c = hl.compress(torch.from_numpy(fields[i]).cuda(), 1e-3, np.inf, mg.REL)
                    v = hl.decompress(c).cpu().numpy()
                    c = c.cpu().numpy()
                assert _same_container(hl, c, expect[i]), "container differs (thread %d, iteration %d)" % (i, it)
                assert float(np.max(np.abs(v - fields[i]))) <= 1e-3 * float(np.max(np.abs(fields[i])))
            hl.release_cache()
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_decompress_out_is_validated():
    torch, mg, hl = _mods()
    u = smooth_field((20, 30, 40), np.float32)
    c = hl.compress(u, 1e-3, np.inf, mg.REL)
    with pytest.raises(ValueError):
        hl.decompress(c, out=np.zeros((20, 30, 41), np.float32))
    with pytest.raises(ValueError):
        hl.decompress(c, out=np.zeros((20, 30, 40), np.float64))
    cd = torch.from_numpy(c).cuda()
    with pytest.raises(ValueError):
        hl.decompress(cd, out=torch.zeros(20 * 30 * 40 + 1, device="cuda"))
    with pytest.raises(ValueError):
        hl.decompress(cd, out=torch.zeros((20, 30, 40), dtype=torch.float64, device="cuda"))
    ctx = hl.Lossless()
    with pytest.raises(ValueError):
        ctx.decompress(torch.zeros(64, dtype=torch.int32, device="cuda"), 10)
    with pytest.raises(ValueError):
        ctx.decompress(torch.zeros((8, 8), dtype=torch.uint8, device="cuda"), 10)
    ctx.close()


@pytest.mark.parametrize("lossless", ["HUFFMAN"])
def test_decoder_follows_the_arriving_record(lossless, monkeypatch):
    """A record of 32 MB and more in host memory is decoded piece by piece while it travels (the
    decoder's launches follow the pieces of the copy on another stream; MGH_HL_DECODE_FOLLOWS=0: copy,
    then decode): same reconstruction from pageable and from registered memory, with the symbol
    widths both paths use."""
    torch, mg, hl = _mods()
    shape = (300, 400, 512)
    u = smooth_field(shape, np.float32)
    c = hl.compress(u, 1e-3, np.inf, mg.REL)
    assert c.size > (40 << 20)
    monkeypatch.setenv("MGH_HL_DECODE_FOLLOWS", "0")
    ref = hl.decompress(c)
    assert float(np.max(np.abs(ref - u))) <= 1e-3 * float(np.max(np.abs(u)))
    monkeypatch.setenv("MGH_HL_DECODE_FOLLOWS", "1")
    for sym16 in ("0", "1"):
        monkeypatch.setenv("MGH_SYM16_DECODE", sym16)
        assert np.array_equal(hl.decompress(c), ref)
        cp = c.copy()
        hl.pin(cp)
        try:
            assert np.array_equal(hl.decompress(cp), ref)
        finally:
            hl.unpin(cp)
    monkeypatch.delenv("MGH_SYM16_DECODE")
    # a record truncated in the middle of its code units: an error, not a hang or a fault
    with pytest.raises(mg.MgardHipError):
        hl.decompress(c[:c.size // 2])
