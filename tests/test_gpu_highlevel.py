"""GPU tests of the high-level path (include/mgard_hip_compress.h): lossless stage, container,
domain decomposition and the whole-array compress / decompress round trip, all through the C
ABI. The serialized records are additionally read back by an independent pure-Python reader
(tests/payload.py) written from the format description."""
import numpy as np
import pytest

import oracle
from tests import payload as pl
from tests.util import nonuniform_coords, smooth_field

pytestmark = pytest.mark.gpu


def _mods():
    import torch
    import mgard_amd
    from mgard_amd import highlevel as hl
    return torch, mgard_amd, hl


def _symbols(n, seed=3, width=6.0, dict_size=8192):
    rng = np.random.default_rng(seed)
    q = np.rint(rng.normal(0, width, n)).astype(np.int64) + dict_size // 2
    return np.clip(q, 0, dict_size - 1)


@pytest.mark.parametrize("lossless", ["HUFFMAN", "HUFFMAN_ZSTD"])
@pytest.mark.parametrize("n,chunk", [(1, 20480), (777, 64), (20480, 20480), (20481, 20480), (300001, 4096)])
def test_lossless_roundtrip_and_record_layout(n, chunk, lossless):
    torch, mg, hl = _mods()
    mode = getattr(hl, lossless)
    q = _symbols(n)
    oi = np.array([0, n // 2, n - 1][:min(3, n)], dtype=np.int64)
    oi = np.unique(oi)
    ov = np.array([-70000, 123456, 8192][:len(oi)], dtype=np.int64)
    q[oi] = 0
    ctx = hl.Lossless()
    qd = torch.from_numpy(q).cuda()
    rec = ctx.compress(qd, 8192, chunk, mode, 3, torch.from_numpy(oi).cuda(), torch.from_numpy(ov).cuda())
    if mode == hl.HUFFMAN:
        r = pl.parse_huffman_record(rec)
        assert r["primary_count"] == n and r["dict_size"] == 8192 and r["chunk_size"] == chunk
        nchunk = (n - 1) // chunk + 1
        assert len(r["bits"]) == nchunk
        words = (r["bits"].astype(np.int64) + 63) // 64
        assert np.array_equal(r["entry"], np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.uint64))
        assert len(r["units"]) == int(words.sum())
        assert np.array_equal(r["outlier_idx"].astype(np.int64), oi) and np.array_equal(r["outliers"], ov)
        if n <= 30000:
            assert np.array_equal(pl.decode_huffman_record(r), q)   # independent decoder
    back, bi, bv = ctx.decompress(rec, n, mode)
    assert np.array_equal(back.cpu().numpy(), q)
    assert np.array_equal(bi.cpu().numpy(), oi) and np.array_equal(bv.cpu().numpy(), ov)
    ctx.close()


def test_lossless_single_symbol_and_full_dictionary():
    torch, mg, hl = _mods()
    ctx = hl.Lossless()
    for q in (np.full(5000, 4096, np.int64), np.arange(8192, dtype=np.int64).repeat(3)):
        rec = ctx.compress(torch.from_numpy(q).cuda(), 8192, 1024)
        assert np.array_equal(pl.decode_huffman_record(pl.parse_huffman_record(rec)), q)
        back, _, _ = ctx.decompress(rec, q.size)
        assert np.array_equal(back.cpu().numpy(), q)
    ctx.close()


@pytest.mark.parametrize("kind", ["wide", "skewed"])
def test_lossless_long_codes_and_many_chunks(kind):
    """Thousands of chunks (unit offsets by look-back across workgroups) and codes longer than
    the decoder's root table (second-level tables) / than root + 11 bits (comparison path)."""
    torch, mg, hl = _mods()
    rng = np.random.default_rng(11)
    if kind == "wide":
        q = _symbols(3_000_000, seed=5, width=900.0)
    else:  # Fibonacci frequencies: the deepest Huffman tree a histogram of this size can give
        fib = [1, 1]
        while len(fib) < 31:
            fib.append(fib[-1] + fib[-2])
        parts = [np.full(f, 100 + k, np.int64) for k, f in enumerate(fib)]
        q = rng.permutation(np.concatenate(parts))
    code = np.zeros(8192, np.uint64)
    freq = np.bincount(q, minlength=8192).astype(np.uint32)
    code, first, entry, keys = hl.huffman_codebook(freq)
    lengths = (code >> np.uint64(56)).astype(np.int64)
    assert lengths.max() > (23 if kind == "skewed" else 12)
    ctx = hl.Lossless()
    for chunk in (1024, 20480):
        rec = ctx.compress(torch.from_numpy(q).cuda(), 8192, chunk)
        r = pl.parse_huffman_record(rec)
        words = (r["bits"].astype(np.int64) + 63) // 64
        assert np.array_equal(r["entry"], np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.uint64))
        assert len(r["units"]) == int(words.sum())
        assert int(r["bits"].sum()) == int(lengths[q].sum())     # every symbol with its code length
        back, _, _ = ctx.decompress(rec, q.size)
        assert np.array_equal(back.cpu().numpy(), q)
    small = q[:40000]
    rec = ctx.compress(torch.from_numpy(small).cuda(), 8192, 2048)
    assert np.array_equal(pl.decode_huffman_record(pl.parse_huffman_record(rec)), small)
    ctx.close()


def test_lossless_rejects_damaged_records():
    torch, mg, hl = _mods()
    ctx = hl.Lossless()
    q = _symbols(50000)
    rec = bytearray(ctx.compress(torch.from_numpy(q).cuda(), 8192, 4096))
    with pytest.raises(mg.MgardHipError):
        ctx.decompress(bytes(rec[:100]), q.size)
    with pytest.raises(mg.MgardHipError):
        ctx.decompress(bytes(rec), q.size + 1)
    bad = bytearray(rec)
    bad[24 + 8 * 13 + 8 * 13 - 1] = 0x7f   # a chunk entry far outside the code stream
    with pytest.raises(mg.MgardHipError):
        ctx.decompress(bytes(bad), q.size)
    ctx.close()


def _err(u, v, s, shape):
    if np.isinf(s):
        return float(np.max(np.abs(u.astype(np.float64) - v.astype(np.float64))))
    return float(np.sqrt(np.mean((u.astype(np.float64) - v.astype(np.float64)) ** 2)))


def _norm(u, s):
    return float(np.max(np.abs(u))) if np.isinf(s) else float(np.sqrt(np.mean(u.astype(np.float64) ** 2)))


@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape,s,mode,lossless", [
    ((65, 70, 129), np.inf, "REL", "HUFFMAN"), ((100, 36, 260), 0.0, "REL", "HUFFMAN_ZSTD"),
    ((1025,), np.inf, "ABS", "HUFFMAN"), ((129, 200), 1.0, "ABS", "HUFFMAN"),
    ((6, 20, 17, 33), np.inf, "REL", "HUFFMAN"), ((3, 4, 5, 6, 7), np.inf, "ABS", "HUFFMAN_ZSTD"),
    # (round 6: a short fastest extent -- 64 x 4 tiles --, the XGC data set's aspect -- a long r-dimension
    # over a small cross-section, chunked solves of the slices --, D = 5 with a REL bound -- norm and
    # quantizer table resident on the device)
    ((70, 300, 9), np.inf, "REL", "HUFFMAN"), ((8, 70, 200, 5), np.inf, "REL", "HUFFMAN"),
    ((5, 2300, 19, 19), np.inf, "REL", "HUFFMAN"), ((5, 5, 20, 20, 40), 0.0, "REL", "HUFFMAN")])
def test_compress_decompress_roundtrip(shape, s, mode, lossless, dt, where):
    torch, mg, hl = _mods()
    u = smooth_field(shape, dt)
    tol = 1e-3
    cfg = hl.Config(lossless=getattr(hl, lossless))
    src = u if where == "host" else torch.from_numpy(u).cuda()
    buf = hl.compress(src, tol, s, mg.REL if mode == "REL" else mg.ABS, config=cfg)
    raw = buf if where == "host" else buf.cpu().numpy()
    m = hl.metadata_parse(bytes(raw[:4096]) if raw.size > 4096 else bytes(raw))
    assert m["shape"] == list(shape) and m["tol"] == tol and m["lossless"] == getattr(hl, lossless)
    assert m["domain_decomposed"] is False and m["dict_size"] == 8192 and m["block_size"] == 20480
    if mode == "REL":
        assert abs(m["norm"] - _norm(u, s)) <= 1e-5 * _norm(u, s)
        if np.isinf(s):
            assert dt(m["norm"]) == dt(np.max(np.abs(u)))
    if u.nbytes > (1 << 20) and min(shape[-3:]) >= 32:
        assert raw.size < u.nbytes                  # it does compress (the 64 KiB decodebook
                                                    # makes tiny arrays fall back to raw storage, and
                                                    # three periods across nine nodes are not smooth)
    assert hl.infer(buf) == (tuple(shape), hl.DOUBLE if dt == np.float64 else hl.FLOAT)
    v = hl.decompress(buf)
    v = v if where == "host" else v.cpu().numpy()
    assert v.shape == tuple(shape) and v.dtype == dt
    bound = tol * (_norm(u, s) if mode == "REL" else 1.0)
    assert _err(u, v, s, shape) <= bound * (1 + 1e-6)


@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("shape,dt", [((70, 300, 9), np.float32), ((64, 70, 200, 5), np.float32),
                                      ((5, 2300, 39, 39), np.float64), ((6, 6, 40, 40, 48), np.float32),
                                      ((2500, 40, 48), np.float64), ((40, 40, 40, 40), np.float32)])
def test_roundtrip_on_the_shapes_of_round_6(shape, dt, where):
    """A short fastest extent (64 x 4 tiles), many t-slices (one launch per level and kind of slice on
    the way back), the XGC data set's aspect (long r over a small cross-section: chunked solves, march
    class by slices), D = 5 (norm and quantizer table resident on the device), long strided pencils in
    chunks -- on a field that compresses, so that the way back runs the kernels and not the raw copy."""
    torch, mg, hl = _mods()
    from tests.util import inside_field
    u = inside_field(shape, dt)
    src = u if where == "host" else torch.from_numpy(u).cuda()
    buf = hl.compress(src, 1e-3, np.inf, mg.REL)
    raw = buf if where == "host" else buf.cpu().numpy()
    assert raw.size < 0.7 * u.nbytes, "the field must compress for this test to mean anything"
    v = hl.decompress(buf)
    v = v if where == "host" else v.cpu().numpy()
    assert v.shape == tuple(shape) and v.dtype == dt
    assert float(np.max(np.abs(v.astype(np.float64) - u))) <= 1e-3 * float(np.max(np.abs(u))) * (1 + 1e-6)


def test_device_input_still_being_produced_on_the_default_stream():
    """The pipeline streams are ordered against the NULL stream (reference queues:
    DeviceAdapterHip.h:514): an input that earlier kernels on torch's default stream are still
    producing -- no torch.cuda.synchronize() in between -- must be compressed as it will be, and a
    reconstruction must be complete for work queued behind it."""
    torch, mg, hl = _mods()
    shape = (257, 260, 300)
    u = smooth_field(shape, np.float32)
    base = torch.from_numpy(u).cuda()
    # (the records themselves may differ from run to run: the outlier list is in atomic order)
    ref = hl.decompress(hl.compress(base * 3.0 + 1.0, 1e-3, np.inf, mg.REL)).cpu().numpy()
    torch.cuda.synchronize()
    for _ in range(3):
        d = base.clone()
        for _k in range(40):       # a queue of kernels that keeps the default stream busy
            d = d * 1.0
        d = d * 3.0 + 1.0          # the value to compress exists only after all of them
        buf = hl.compress(d, 1e-3, np.inf, mg.REL)
        v = hl.decompress(buf)
        w = (v - d).abs().max()    # queued on the default stream right behind the pipeline
        assert np.array_equal(v.cpu().numpy(), ref)
        assert float(w) <= 1e-3 * float(d.abs().max()) * (1 + 1e-6)


@pytest.mark.parametrize("shape", [(40, 33, 70), (3, 4, 5, 6, 7), (300, 5, 7), (40, 50)],
                         ids=["fused 3-D", "5-D", "thin 3-D", "2-D"])
@pytest.mark.parametrize("mode", ["REL", "ABS"])
def test_all_zero_input(mode, shape):
    """norm == 0: the reference replaces it by epsilon (NormCalculator.hpp:61-66); the header
    must keep the size it was reserved with and the round trip must give zeros back -- on the fused
    path and on the staged one (norm and quantizer table resident on the device there too)."""
    torch, mg, hl = _mods()
    u = np.zeros(shape, np.float32)
    for src in (u, torch.from_numpy(u).cuda()):
        buf = hl.compress(src, 1e-3, np.inf, mg.REL if mode == "REL" else mg.ABS)
        v = hl.decompress(buf)
        v = v if isinstance(v, np.ndarray) else v.cpu().numpy()
        assert v.shape == u.shape and not v.any()


def test_outlier_estimate_too_small_is_retried():
    """estimate_outlier_ratio sizes the outlier lists; when a subdomain has more outliers than
    that, the reference re-allocates and re-launches the quantizer (LinearQuantization.hpp:
    621-676). Same here: the call succeeds and the stream equals the generously sized one."""
    torch, mg, hl = _mods()
    rng = np.random.default_rng(5)
    u = (smooth_field((64, 65, 66), np.float32) + rng.normal(0, 30.0, (64, 65, 66)).astype(np.float32) *
         (rng.random((64, 65, 66)) < 0.02))          # 2 % spikes: far outside a 64-entry dictionary
    small = hl.Config(estimate_outlier_ratio=1e-5, huff_dict_size=64)
    big = hl.Config(estimate_outlier_ratio=1.0, huff_dict_size=64)
    for src in (u, torch.from_numpy(u).cuda()):
        a = hl.decompress(hl.compress(src, 1e-4, np.inf, mg.REL, config=small))
        b = hl.decompress(hl.compress(src, 1e-4, np.inf, mg.REL, config=big))
        a = a if isinstance(a, np.ndarray) else a.cpu().numpy()
        b = b if isinstance(b, np.ndarray) else b.cpu().numpy()
        assert np.array_equal(a, b)
        assert float(np.max(np.abs(a - u))) <= 1e-4 * float(np.max(np.abs(u))) * (1 + 1e-6)


def _reference_footprint(shape, elem, ratio=1.0, dict_size=8192, block=20480, prefetch=False):
    """DomainDecomposer::EstimateMemoryFootprint (DomainDecomposer.hpp:24-69 and the estimators it
    calls), runtime-independent terms -- restated here independently of highlevel.hip."""
    D = len(shape)
    n = float(np.prod(shape, dtype=np.float64))
    ws = float(np.prod([e + 2 for e in shape], dtype=np.float64))
    def levels(e):
        k = 0
        while e > 2:
            e = e // 2 + 1
            k += 1
        return k
    L = min(levels(e) for e in shape)
    hier = 0.0
    for l in range(L + 1):
        for e in shape:
            m = e
            for _ in range(L - l):
                m = m // 2 + 1
            hier += 6.0 * (m + 1) * elem
        hier += D * 8 * 2
    b = n * elem + n * 8 + ratio * 8 + hier
    if prefetch:
        b *= 2
    nchunk = np.floor((n - 1) / block) + 1
    lossless = (8 + n * ratio * 16 + dict_size * 4 + dict_size * 8 + (8 * 128 + 8 * dict_size) + n * 8 +
                3 * nchunk * 8 + 4 + dict_size * 4 + dict_size * 8 + 16 * dict_size + 24 * dict_size +
                8 * dict_size + 64)
    comp = ws * elem * (2 if D > 3 else 1) + elem + (L + 1) * elem + lossless + elem
    if 8 > elem:
        comp += 8 * n
    return int(b + comp)


def test_maxdim_split_follows_the_reference_footprint():
    """The auto-split happens exactly where the reference's estimate reaches the budget
    (need = estimate >= available, DomainDecomposer.hpp:72-88), and halves the largest dimension
    (rounding up) until the estimate with prefetch fits (:199-228)."""
    torch, mg, hl = _mods()
    shape = (66, 300, 80)
    u = smooth_field(shape, np.float32)
    est = _reference_footprint(shape, 4)
    assert 45 * u.size < est < 52 * u.size          # ~48 bytes per element for float32 (no prefetch)
    m = hl.metadata_parse(bytes(hl.compress(u, 1e-2, np.inf, mg.REL, config=hl.Config(max_memory_footprint=est + 1))))
    assert not m["domain_decomposed"]
    m = hl.metadata_parse(bytes(hl.compress(u, 1e-2, np.inf, mg.REL, config=hl.Config(max_memory_footprint=est))))
    assert m["domain_decomposed"] and m["dd_method"] == hl.DD_MAXDIM and m["dd_dim"] == 1 and m["dd_size"] == 150
    # a budget the halves (with prefetch) do not fit: quarters
    half = _reference_footprint((66, 150, 80), 4, prefetch=True)
    m = hl.metadata_parse(bytes(hl.compress(u, 1e-2, np.inf, mg.REL, config=hl.Config(max_memory_footprint=half))))
    assert m["dd_size"] == 75


def test_container_records_hold_the_quantized_coefficients():
    """The single record of a non-decomposed stream decodes (independent reader) to exactly the
    integers the low-level path produces, outliers included."""
    torch, mg, hl = _mods()
    shape = (65, 40, 65)
    u = smooth_field(shape, np.float32, noise=2e-3)
    tol = 1e-2
    buf = hl.compress(u, tol, np.inf, mg.REL, config=hl.Config(huff_dict_size=2048, huff_block_size=4096))
    m = hl.metadata_parse(bytes(buf))
    assert m["dict_size"] == 2048 and m["block_size"] == 4096
    recs = pl.split_container(buf, m["metadata_size"])
    assert len(recs) == 1 and len(recs[0]) < u.nbytes
    r = pl.parse_huffman_record(recs[0])
    h = mg.Hierarchy(shape, np.float32)
    q, oi, ov, cnt, nrm = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.REL, tol, np.inf,
                                               dict_size=2048)
    assert cnt > 0
    assert np.array_equal(pl.decode_huffman_record(r), q.cpu().numpy().reshape(-1))
    a = np.argsort(r["outlier_idx"])
    b = np.argsort(oi.cpu().numpy())
    assert np.array_equal(r["outlier_idx"][a].astype(np.int64), oi.cpu().numpy()[b])
    assert np.array_equal(r["outliers"][a], ov.cpu().numpy()[b])
    h.close()


@pytest.mark.parametrize("s", [np.inf, 0.0])
@pytest.mark.parametrize("kind", ["maxdim_auto", "block", "variable"])
def test_domain_decomposition(kind, s):
    torch, mg, hl = _mods()
    shape = (66, 300, 80)
    u = smooth_field(shape, np.float32)
    tol = 1e-2
    if kind == "maxdim_auto":
        # a memory budget that only fits a fraction of the array forces a MaxDim split
        # (the reference plans ~48 bytes per float32 element, test_maxdim_split_follows_...)
        cfg = hl.Config(max_memory_footprint=40 * u.size)
    elif kind == "block":
        cfg = hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=50)
    else:
        cfg = hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=1,
                        domain_decomposition_sizes=[100, 60, 140])
    buf = hl.compress(u, tol, s, mg.REL, config=cfg)
    m = hl.metadata_parse(bytes(buf))
    assert m["domain_decomposed"]
    recs = pl.split_container(buf, m["metadata_size"])
    if kind == "maxdim_auto":
        assert m["dd_method"] == hl.DD_MAXDIM and m["dd_dim"] == 1
        assert len(recs) == (300 - 1) // m["dd_size"] + 1 and len(recs) > 1
    elif kind == "block":
        assert m["dd_method"] == hl.DD_BLOCK and m["dd_size"] == 50 and len(recs) == 2 * 6 * 2
    else:
        assert m["dd_method"] == hl.DD_VARIABLE and len(recs) == 3
    nrm = _norm(u, s)
    assert abs(m["norm"] - nrm) <= 1e-5 * nrm
    v = hl.decompress(buf, config=cfg)
    assert _err(u, v, s, shape) <= tol * nrm * (1 + 1e-6)
    # subdomain extents in record order (last decomposed dimension fastest,
    # DomainDecomposer.hpp:105-113); every record is the Huffman record of its subdomain, or the
    # raw subdomain where that is smaller (GPUPipelines.hpp:136-155)
    def pieces(n, size):
        return [size] * (n // size) + ([n % size] if n % size else [])
    if kind == "maxdim_auto":
        sizes = [66 * p * 80 for p in pieces(300, m["dd_size"])]
    elif kind == "block":
        sizes = [a * b * c for a in pieces(66, 50) for b in pieces(300, 50) for c in pieces(80, 50)]
    else:
        sizes = [66 * p * 80 for p in (100, 60, 140)]
    assert len(sizes) == len(recs) and sum(sizes) == u.size
    n_huff = 0
    for r, n in zip(recs, sizes):
        if len(r) == 4 * n:
            continue
        assert len(r) < 4 * n and pl.parse_huffman_record(r)["primary_count"] == n
        n_huff += 1
    assert n_huff > 0


def test_nonuniform_coordinates_roundtrip():
    torch, mg, hl = _mods()
    shape = (33, 50, 65)
    u = smooth_field(shape, np.float64)
    coords = nonuniform_coords(shape, np.float64)
    buf = hl.compress(u, 1e-4, 0.0, mg.ABS, coords=coords)
    m = hl.metadata_parse(bytes(buf))
    assert not m["uniform"]
    for got, ref in zip(m["coords"], coords):
        assert np.array_equal(got, ref)
    v = hl.decompress(buf)
    # same integers as the low-level path on the same hierarchy => same reconstruction
    h = mg.Hierarchy(shape, np.float64, coords=coords)
    q, oi, ov, cnt, _ = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.ABS, 1e-4, 0.0, norm=1.0)
    w = h.dequantize_recompose(q, mg.ABS, 1e-4, 0.0, 1.0, outlier_idx=oi, outlier_val=ov).cpu().numpy()
    assert np.array_equal(v, w)
    h.close()


@pytest.mark.parametrize("dict_size,block", [(4096, 1000), (16384, 3333), (8192, 70000), (2048, 2000)])
@pytest.mark.parametrize("where", ["host", "device"])
def test_huffman_parameters_roundtrip(dict_size, block, where):
    """Chunk sizes that are not multiples of 8 / below the parallel decoder's limit / beyond the
    encoder's LDS budget (two-pass fallback), small and large dictionaries; the record must hold
    exactly the low-level integers."""
    torch, mg, hl = _mods()
    shape = (65, 97, 130)
    u = smooth_field(shape, np.float32)
    x = u if where == "host" else torch.from_numpy(u).cuda()
    cfg = hl.Config(huff_dict_size=dict_size, huff_block_size=block)
    buf = hl.compress(x, 1e-2, np.inf, mg.REL, config=cfg)
    raw = buf if where == "host" else buf.cpu().numpy()
    m = hl.metadata_parse(bytes(raw[:4096]))
    recs = pl.split_container(raw, m["metadata_size"])
    assert len(recs) == 1 and len(recs[0]) < u.nbytes        # (not the raw fallback)
    r = pl.parse_huffman_record(recs[0])
    assert r["dict_size"] == dict_size and r["chunk_size"] == block
    h = mg.Hierarchy(shape, np.float32)
    q, oi, ov, cnt, _ = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.REL, 1e-2, np.inf, dict_size=dict_size)
    h.close()
    assert np.array_equal(pl.decode_huffman_record(r), q.cpu().numpy().ravel())
    assert sorted(r["outlier_idx"].tolist()) == sorted(oi.cpu().numpy().tolist())
    v = hl.decompress(buf)
    v = v if where == "host" else v.cpu().numpy()
    assert np.max(np.abs(v - u)) <= 1e-2 * np.max(np.abs(u))
    hl.release_cache()


def test_alternative_kernels_give_the_same_result():
    """The developer switches select the previous / simpler kernels (one fine row per wave in the
    node restore, every level with its own launches, the parallel decoder without rings, 16-bit
    symbols on the decompression side): same container size, bit-identical reconstruction."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "crosscheck_worker.py")

    def run(extra):
        env = dict(os.environ)
        env.update(extra)
        r = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGESTS ")]
        assert line, r.stdout[-2000:]
        return line[-1]

    ref = run({})
    assert run({"MGH_RESTORE_ROWS": "1", "MGH_NO_RECOMPOSE_HEAD": "1", "MGH_HUFF_PAR_DECODE": "1"}) == ref
    assert run({"MGH_SYM16_DECODE": "1", "MGH_HUFF_SERIAL_DECODE": "1"}) == ref
    # int64 between decoder and dequantizer / 16-bit symbols on every level (not only the finest)
    assert run({"MGH_SYM16_DECODE": "0"}) == ref
    assert run({"MGH_SYM16_MIXED": "0"}) == ref
    assert run({"MGH_FORCE_V1": "1"}) == ref
    # the round-2 level kernel's tile shapes / chunk lengths / variants and the Thomas solvers
    assert run({"MGH_FUSED_WIDE": "0", "MGH_FUSED_FIXED": "0", "MGH_RCH": "2,3,8"}) == ref
    assert run({"MGH_FUSED_WIDE": "2", "MGH_FUSED_XCD": "0", "MGH_FUSED_FACES": "0", "MGH_IPK_STREAM": "0"}) == ref
    assert run({"MGH_FUSED_WIDE": "0", "MGH_FUSED4": "0", "MGH_IPK_W": "32"}) == ref
    # round 3: no box kernel / box kernel on every level, tail kernel without the solves of the
    # level above it, other residency plans of the streaming Thomas solves
    assert run({"MGH_BOX": "0", "MGH_TAIL_SOLVES": "0", "MGH_RESTORE_V": "2", "MGH_IPK_CONTIG": "2"}) == ref
    assert run({"MGH_BOX": "3", "MGH_IPK_WPC": "16"}) == ref
    # round 6: tile placement and shapes, slice batching, chunked strided solves, the N-D row kernels
    assert run({"MGH_FUSED_XCD": "2", "MGH_FUSED_TALL": "0", "MGH_SLICE_BATCH": "0"}) == ref
    assert run({"MGH_IPK_SPEC_LONG": "0", "MGH_ND_ROWS": "0", "MGH_FUSED_XCD": "0"}) == ref


def test_incompressible_subdomain_is_stored_raw():
    torch, mg, hl = _mods()
    rng = np.random.default_rng(1)
    for shape in [(40, 41, 42), (5, 12, 13, 14)]:
        u = rng.standard_normal(shape).astype(np.float32) * 1e6
        buf = hl.compress(u, 1e-9, np.inf, mg.ABS)
        m = hl.metadata_parse(bytes(buf))
        recs = pl.split_container(buf, m["metadata_size"])
        assert len(recs) == 1 and len(recs[0]) == u.nbytes      # GPUPipelines.hpp:136-155
        assert np.array_equal(np.frombuffer(recs[0], np.float32).reshape(shape), u)
        assert np.array_equal(hl.decompress(buf), u)


def test_output_too_large_and_bad_arguments():
    torch, mg, hl = _mods()
    u = smooth_field((65, 65, 65), np.float32)
    with pytest.raises(mg.MgardHipError, match="-7"):
        hl.compress(u, 1e-6, np.inf, mg.REL, out_capacity=2000)
    with pytest.raises(mg.MgardHipError):
        hl.compress(u, 1e-3, np.inf, mg.REL, config=hl.Config(lossless=hl.HUFFMAN_LZ4))
    with pytest.raises(mg.MgardHipError):
        hl.compress(smooth_field((2, 65, 65), np.float32), 1e-3)
    buf = hl.compress(u, 1e-3, np.inf, mg.REL)
    bad = buf.copy()
    bad[len(bad) // 2:] = 0
    with pytest.raises(mg.MgardHipError):
        hl.decompress(bad[:len(bad) - 40])
    hl.release_cache()


def test_matches_reference_error_semantics_at_benchmark_size():
    """512^3 float32, REL 1e-3, s = inf, device-resident: end-to-end compress + decompress through
    the container; the error bound holds and the stream is much smaller than the input."""
    torch, mg, hl = _mods()
    u = smooth_field((512, 512, 512), np.float32)
    ud = torch.from_numpy(u).cuda()
    buf = hl.compress(ud, 1e-3, np.inf, mg.REL)
    assert buf.numel() < u.nbytes / 3     # (the 1e-3 noise of the synthetic field is incompressible)
    v = hl.decompress(buf)
    nrm = float(np.max(np.abs(u)))
    assert float((v - ud).abs().max().item()) <= 1e-3 * nrm


def test_decompress_survives_random_damage():
    """Bit flips anywhere in a valid stream (header, chunk tables, decodebook, code units, outlier
    lists, record sizes): mgh_decompress either reports an error or returns an array -- it never
    crashes, hangs or reads outside its buffers."""
    torch, mg, hl = _mods()
    u = smooth_field((40, 50, 66), np.float32)
    cfg = hl.Config(huff_dict_size=1024, huff_block_size=1024, domain_decomposition=hl.DD_BLOCK,
                    block_size=33)
    good = hl.compress(u, 1e-2, np.inf, mg.REL, config=cfg)
    assert _err(u, hl.decompress(good, config=cfg), np.inf, u.shape) <= 1e-2 * float(np.max(np.abs(u))) * 1.000001
    rng = np.random.default_rng(11)
    outcomes = {"error": 0, "array": 0}
    m = hl.metadata_parse(bytes(good))
    for trial in range(150):
        bad = good.copy()
        if trial % 3 == 0:      # damage the structured front part of a record
            pos = m["metadata_size"] + int(rng.integers(0, 9000))
        elif trial % 3 == 1:    # ... or its tail (outlier count / indices / values)
            pos = bad.size - 1 - int(rng.integers(0, 4000))
        else:
            pos = int(rng.integers(0, bad.size))
        bad[pos % bad.size] ^= np.uint8(1 << int(rng.integers(0, 8)))
        if trial % 7 == 0:      # a burst of garbage
            a = int(rng.integers(0, bad.size - 64))
            bad[a:a + 64] = rng.integers(0, 256, 64, dtype=np.uint8)
        if trial % 10 == 9:
            bad = bad[:int(rng.integers(20, bad.size))]
        try:
            v = hl.decompress(bad, config=cfg)
            assert v.shape == u.shape
            outcomes["array"] += 1
        except mg.MgardHipError:
            outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["array"] > 0
    hl.release_cache()


@pytest.mark.parametrize("shape,dt,s,mode,ndev", [
    ((40, 65, 70), np.float32, np.inf, "REL", 2), ((37, 33, 50), np.float64, 0.0, "REL", 3),
    ((64, 20, 33), np.float32, np.inf, "ABS", 4), ((24, 9, 17, 33), np.float32, np.inf, "REL", 2),
    ((5, 40, 40), np.float32, np.inf, "REL", 4)])
def test_multi_device_api_on_one_gpu(shape, dt, s, mode, ndev):
    """mgh_compress_multi / mgh_decompress_multi (one process, one host thread per device): with
    device 0 listed several times the whole control flow -- slabs of the slowest dimension, norm
    of the whole domain, per-slab ABS bound, records framed in id order behind one header -- runs
    on a single GPU. The container must be the one mgh_compress writes for the same MaxDim
    decomposition: both decoders must read both containers, with identical reconstructions."""
    torch, mg, hl = _mods()
    u = smooth_field(shape, dt)
    m = mg.REL if mode == "REL" else mg.ABS
    devs = (0,) * ndev
    buf = hl.compress_multi(u, 1e-3, s, m, devices=devs)
    meta = hl.metadata_parse(bytes(buf[:4096]) if buf.size > 4096 else bytes(buf))
    assert meta["shape"] == list(shape)
    if shape[0] >= 6:
        assert meta["domain_decomposed"] is True
    if mode == "REL":
        assert abs(meta["norm"] - _norm(u, s)) <= 1e-5 * _norm(u, s)
    v1 = hl.decompress_multi(buf, devices=devs)
    v2 = hl.decompress(buf)                      # the single-device reader on the same stream
    assert np.array_equal(v1, v2)
    bound = 1e-3 * (_norm(u, s) if mode == "REL" else 1.0)
    assert _err(u, v1, s, shape) <= bound * (1 + 1e-6)
    v3 = hl.decompress_multi(hl.compress(u, 1e-3, s, m), devices=devs)   # not decomposed: falls back
    assert _err(u, v3, s, shape) <= bound * (1 + 1e-6)


@pytest.mark.parametrize("shape,dt,ndev,mode,s", [((24, 33, 40), np.float32, 3, "REL", np.inf),
                                                  ((10, 20, 130), np.float64, 2, "REL", 0.0),
                                                  ((16, 9, 17, 20), np.float32, 2, "ABS", np.inf)])
def test_multi_device_api_device_resident_input(shape, dt, ndev, mode, s):
    """mgh_compress_multi with a device-resident volume (GPUPipelines.hpp:69-207 takes device
    pointers): slabs of the source device are compressed where they are, the others would travel by
    hipMemcpyPeerAsync (here every "device" is device 0: the same code path with a same-device
    peer copy is not taken -- the in-place path is). The container comes back in device memory and
    is byte for byte the one the host-buffer call writes."""
    torch, mg, hl = _mods()
    u = smooth_field(shape, dt)
    m = mg.REL if mode == "REL" else mg.ABS
    devs = (0,) * ndev
    ref = hl.compress_multi(u, 1e-3, s, m, devices=devs)
    got = hl.compress_multi(torch.from_numpy(u).cuda(), 1e-3, s, m, devices=devs)
    assert got.is_cuda
    g = got.cpu().numpy()
    if mode == "REL" and s != np.inf:
        # (the L2 norm of the header is a sum whose order differs between the host-staged and the
        # in-place slab: the payloads may differ in the last bits; both must decode within the bound)
        v = hl.decompress(g)
        assert _err(u, v, s, shape) <= 1e-3 * _norm(u, s) * (1 + 1e-6)
    else:
        assert np.array_equal(g, ref)
    v1 = hl.decompress_multi(g, devices=devs)
    bound = 1e-3 * (_norm(u, s) if mode == "REL" else 1.0)
    assert _err(u, v1, s, shape) <= bound * (1 + 1e-6)


def test_multi_device_peer_copy_path_on_one_gpu():
    """The slab scatter of a device-resident volume (hipMemcpyPeerAsync into the worker's buffer,
    one copy serving norm and compression) forced on although every slab is already local
    (MGH_MULTI_FORCE_PEER=1, read once per process: a child process). Same container."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, torch, hashlib\n"
        "import mgard_amd as mg\n"
        "from mgard_amd import highlevel as hl\n"
        "from tests.util import smooth_field\n"
        "u = smooth_field((24, 33, 40), np.float32)\n"
        "g = hl.compress_multi(torch.from_numpy(u).cuda(), 1e-3, np.inf, mg.REL, devices=(0, 0, 0))\n"
        "print('DIGEST', hashlib.sha256(g.cpu().numpy().tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        env = dict(os.environ)
        env.update(extra)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        return [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")][-1]

    assert run({"MGH_MULTI_FORCE_PEER": "1"}) == run({})


@pytest.mark.parametrize("shape,dt,ndev", [((12, 40, 50), np.float64, 2), ((9, 33, 70), np.float32, 3)])
def test_multi_device_api_nonuniform_coordinates(shape, dt, ndev):
    """The slabs of a multi-device compression carry their own part of the non-uniform coordinate
    of the slowest dimension (the reference's subdomain coordinates, DomainDecomposer.hpp:260-303),
    the header carries the whole one: both readers reconstruct the same values within the bound."""
    torch, mg, hl = _mods()
    u = smooth_field(shape, dt)
    coords = nonuniform_coords(shape, dt)
    devs = (0,) * ndev
    buf = hl.compress_multi(u, 1e-3, np.inf, mg.ABS, devices=devs, coords=coords)
    meta = hl.metadata_parse(bytes(buf[:8192]) if buf.size > 8192 else bytes(buf))
    assert meta["shape"] == list(shape) and meta["domain_decomposed"] is True
    v1 = hl.decompress_multi(buf, devices=devs)
    v2 = hl.decompress(buf)
    assert np.array_equal(v1, v2)
    assert float(np.max(np.abs(v1.astype(np.float64) - u.astype(np.float64)))) <= 1e-3
    # the same decomposition written by the single-device path reads back identically through
    # the multi-device reader
    w = hl.decompress_multi(hl.compress(u, 1e-3, np.inf, mg.ABS, coords=coords), devices=devs)
    assert float(np.max(np.abs(w.astype(np.float64) - u.astype(np.float64)))) <= 1e-3


@pytest.mark.parametrize("shape,dt,dict_size,dd", [
    ((65, 70, 129), np.float32, 8192, False), ((33, 40, 36), np.float64, 64, False),
    ((8, 20, 17, 33), np.float32, 8192, False), ((30001,), np.float32, 8192, False),
    ((130, 70, 129), np.float32, 8192, True)])
def test_reorder_1_level_linearised_stream(shape, dt, dict_size, dd):
    """Config::reorder = 1: the lossless stage is fed the integers level by level
    (LinearQuantization.hpp:46-146, 588-605) and the header says so. The record decodes -- with
    the restated reference decoder -- to exactly the oracle's linearisation of the integers the
    ordinary path produces, outliers sit at their linearised positions, and mgh_decompress
    reconstructs the same values as from a reorder = 0 stream."""
    torch, mg, hl = _mods()
    import oracle
    u = smooth_field(shape, dt)
    kw = dict(huff_dict_size=dict_size, reorder=1)
    if dd:
        kw.update(domain_decomposition=hl.DD_MAXDIM, max_memory_footprint=40 * u.size)
    buf = hl.compress(u, 1e-3, np.inf, mg.REL, config=hl.Config(**kw))
    meta = hl.metadata_parse(bytes(buf[:4096]) if buf.size > 4096 else bytes(buf))
    assert meta["reorder"] == 1 and meta["domain_decomposed"] is dd
    if dd:
        assert buf.size < 0.95 * u.nbytes      # (Huffman records, not raw subdomains)
    kw0 = dict(kw, reorder=0)
    buf0 = hl.compress(u, 1e-3, np.inf, mg.REL, config=hl.Config(**kw0))
    assert hl.metadata_parse(bytes(buf0[:4096]) if buf0.size > 4096 else bytes(buf0))["reorder"] == 0
    v1, v0 = hl.decompress(buf), hl.decompress(buf0)
    assert np.array_equal(v1, v0)
    nrm = float(np.max(np.abs(u)))
    assert float(np.max(np.abs(v1.astype(np.float64) - u))) <= 1e-3 * nrm
    if not dd and u.size <= 40000:
        recs = pl.split_container(buf, meta["metadata_size"])
        assert len(recs) == 1 and len(recs[0]) < u.nbytes      # (not stored raw)
        r = pl.parse_huffman_record(recs[0])
        lin = pl.decode_huffman_record(r)
        h = mg.Hierarchy(shape, dt)
        q, oi, ov, cnt, _ = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.REL, 1e-3, np.inf,
                                                 dict_size=dict_size, outlier_cap=u.size)
        o = oracle.Hierarchy(shape, dt)
        assert np.array_equal(lin, o.level_linearize(q.cpu().numpy()))
        want = sorted((o.linearized_position(int(i)), int(v)) for i, v in zip(oi.cpu().numpy()[:cnt], ov.cpu().numpy()[:cnt]))
        got = sorted(zip(r["outlier_idx"].tolist(), r["outliers"].tolist()))
        assert want == got
        if dict_size == 64:
            assert cnt > 50
        h.close()


@pytest.mark.parametrize("writer_reorder,reader_reorder", [(1, 0), (0, 1)])
def test_multi_device_reader_takes_reorder_from_the_stream(writer_reorder, reader_reorder):
    """The slabs of mgh_decompress_multi are decoded with what the STREAM's header says
    (reorder, lossless choice, dictionary), never with the caller's config: a reorder = 1
    container read with the default config -- and the converse -- reconstructs exactly what
    mgh_decompress reconstructs from the same bytes."""
    torch, mg, hl = _mods()
    shape = (24, 33, 40)
    u = smooth_field(shape, np.float32)
    devs = (0, 0, 0)
    buf = hl.compress_multi(u, 1e-3, np.inf, mg.REL, devices=devs, config=hl.Config(reorder=writer_reorder))
    meta = hl.metadata_parse(bytes(buf[:4096]) if buf.size > 4096 else bytes(buf))
    assert meta["reorder"] == writer_reorder and meta["domain_decomposed"] is True
    ref = hl.decompress(buf)
    got = hl.decompress_multi(buf, devices=devs, config=hl.Config(reorder=reader_reorder))
    assert np.array_equal(got, ref)
    assert float(np.max(np.abs(ref.astype(np.float64) - u))) <= 1e-3 * float(np.max(np.abs(u)))
    # the single-device writer with MaxDim on dim 0 produces the same kind of container
    kw = dict(reorder=writer_reorder, domain_decomposition=hl.DD_MAXDIM, max_memory_footprint=30 * u.size)
    buf2 = hl.compress(u, 1e-3, np.inf, mg.REL, config=hl.Config(**kw))
    if hl.metadata_parse(bytes(buf2[:4096]))["domain_decomposed"]:
        assert np.array_equal(hl.decompress_multi(buf2, devices=devs, config=hl.Config(reorder=reader_reorder)),
                              hl.decompress(buf2))


def test_developer_switches_are_validated(monkeypatch):
    """An MGH_* variable with a value outside its range is an error when a hierarchy is created or
    a high-level call starts -- never a silent default. A name the library does not know (a typo,
    another product's variable) is a warning on stderr, once per process, and breaks nothing."""
    torch, mg, hl = _mods()
    u = smooth_field((9, 9, 9), np.float32)
    monkeypatch.setenv("MGH_FUSED_WIDE", "7")
    with pytest.raises(mg.MgardHipError, match="MGH_FUSED_WIDE"):
        mg.Hierarchy((9, 9, 9), np.float32)
    with pytest.raises(mg.MgardHipError, match="MGH_FUSED_WIDE"):
        hl.compress(u, 1e-3, np.inf, mg.REL)
    monkeypatch.delenv("MGH_FUSED_WIDE")
    monkeypatch.setenv("MGH_FUSED_WIDTH", "1")       # (typo of MGH_FUSED_WIDE)
    mg.Hierarchy((9, 9, 9), np.float32).close()
    monkeypatch.delenv("MGH_FUSED_WIDTH")
    monkeypatch.setenv("MGH_RCH", "2,3")
    with pytest.raises(mg.MgardHipError, match="MGH_RCH"):
        mg.Hierarchy((9, 9, 9), np.float32)
    monkeypatch.setenv("MGH_RCH", "2,3,8")
    mg.Hierarchy((9, 9, 9), np.float32).close()


def _records_on_device(torch, stream, metadata_size):
    """[(offset, size)] of the `[u64 size][payload]` records of a device-resident container."""
    out, at, end = [], int(metadata_size), int(stream.numel())
    while at < end:
        size = int(np.frombuffer(stream[at:at + 8].cpu().numpy().tobytes(), dtype="<u8")[0])
        out.append((at + 8, size))
        at += 8 + size
    assert at == end
    return out


def _outlier_tail(rec):
    """Byte offset of outlier_idx[] in a serialized Huffman record held in a device tensor (layout:
    Huffman.hpp:163-239; every field is 8-byte aligned): what lies before it -- counts, chunk
    table, decodebook, code units, outlier_count -- is a function of the integers alone."""
    def u64(at):
        return int(np.frombuffer(rec[at:at + 8].cpu().numpy().tobytes(), dtype="<u8")[0])
    hm = u64(16)
    p = 24 + 8 * hm
    p += 8 + u64(p)          # decodebook
    p += 8 + 8 * u64(p)      # code units
    count = u64(p)
    # (behind the lists, optional: the decoder's synchronisation points, 8 + 256 bytes per chunk)
    assert rec.numel() - (p + 8 + 16 * count) in (0, 8 + 256 * (hm // 2))
    return p + 8, count


def _config3_volume(torch, nt=64):
    """64 x 512^3 f32 on the device: the 3-D field of the metric drifting slowly along dim 0
    (what bench.py --config 4d uses per slab)."""
    base = torch.from_numpy(smooth_field((512, 512, 512), np.float32)).cuda()
    vol = torch.empty((nt, 512, 512, 512), dtype=torch.float32, device="cuda")
    for t in range(nt):
        vol[t] = base * (1.0 + 0.002 * t) + 1e-4 * t
    return vol


def test_config3_whole_volume_on_one_gpu():
    """BASELINE.json configs[3] as ONE volume: 4-D 64 x 512^3 f32 (34 GB, device-resident), split
    on dim 0 into the 8 slabs of 8 x 512^3 that the 8 ranks of the weak-scaling run own
    (`Variable` decomposition, sizes {8} x 8: DomainDecomposer.hpp:199-230,260-303), REL bound
    against the norm of the WHOLE volume (ErrorToleranceCalculator.hpp:69-89,134-155): eight
    records behind one header (GPUPipelines.hpp:189-193), the round trip within tol * norm, and
    every slab's record byte-identical to the record a stand-alone compression of that slab
    writes under the ABS bound tol * norm -- i.e. the integers of a slab do not depend on whether
    it is compressed as a subdomain or on its own rank."""
    torch, mg, hl = _mods()
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~130 GB of device memory")
    vol = _config3_volume(torch)
    nbytes = vol.numel() * 4
    cfg = hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=0,
                    domain_decomposition_sizes=[8] * 8)
    obuf = torch.empty(nbytes // 2, dtype=torch.uint8, device="cuda")
    stream = hl.compress(vol, 1e-3, np.inf, mg.REL, config=cfg, out=obuf)
    meta = hl.metadata_parse(bytes(stream[:8192].cpu().numpy()))
    assert meta["shape"] == [64, 512, 512, 512] and meta["domain_decomposed"] is True
    assert meta["dd_method"] == hl.DD_VARIABLE and meta["dd_dim"] == 0
    nrm = float(vol.abs().max().item())
    assert np.float32(meta["norm"]) == np.float32(nrm)
    recs = _records_on_device(torch, stream, meta["metadata_size"])
    assert len(recs) == 8
    assert stream.numel() < 0.5 * nbytes
    # stand-alone slabs (what a rank of the 8-GPU run compresses after the norm all-reduce)
    abs_tol = float(np.float32(1e-3) * np.float32(meta["norm"]))
    for k in (0, 3, 7):
        alone = hl.compress(vol[8 * k:8 * (k + 1)], abs_tol, np.inf, mg.ABS)
        am = hl.metadata_parse(bytes(alone[:8192].cpu().numpy()))
        (aoff, asize), = _records_on_device(torch, alone, am["metadata_size"])
        off, size = recs[k]
        assert size == asize
        # identical integers = identical Huffman part of the record (counts, code book, code
        # stream); the outlier list is in atomic order and is compared as a set
        ra, rb = stream[off:off + size].clone(), alone[aoff:aoff + asize].clone()
        (tail_a, na), (tail_b, nb) = _outlier_tail(ra), _outlier_tail(rb)
        assert tail_a == tail_b and na == nb
        assert torch.equal(ra[:tail_a], rb[:tail_b]), k
        assert torch.equal(ra[tail_a + 16 * na:], rb[tail_b + 16 * na:]), k   # synchronisation points
        ia, va = ra[tail_a:tail_a + 8 * na].view(torch.int64), ra[tail_a + 8 * na:tail_a + 16 * na].view(torch.int64)
        ib, vb = rb[tail_b:tail_b + 8 * na].view(torch.int64), rb[tail_b + 8 * na:tail_b + 16 * na].view(torch.int64)
        oa, ob = torch.argsort(ia), torch.argsort(ib)
        assert torch.equal(ia[oa], ib[ob]) and torch.equal(va[oa], vb[ob]), k
        del alone, ra, rb
    back = torch.empty_like(vol)
    # (a Variable decomposition is not recorded in the header -- reference and here alike: the
    # reader is given the sizes the writer used)
    hl.decompress(stream, out=back, config=cfg)
    err = max(float((back[t] - vol[t]).abs().max().item()) for t in range(64))
    assert err <= 1e-3 * nrm
    del back, vol, obuf
    hl.release_cache()
    torch.cuda.empty_cache()


def test_config3_whole_volume_through_the_multi_device_api():
    """The same volume through mgh_compress_multi / mgh_decompress_multi (one process, one host
    thread per listed device; device 0 listed eight times on a one-GPU box): host buffers in and
    out, 8 slabs of 8 x 512^3 behind one MaxDim header that the single-device reader opens too."""
    torch, mg, hl = _mods()
    import psutil
    if psutil.virtual_memory().available < 160e9:
        pytest.skip("needs ~110 GB of host memory")
    vol = _config3_volume(torch)
    nrm = float(vol.abs().max().item())
    u = vol.cpu().numpy()
    del vol
    torch.cuda.empty_cache()
    devs = (0,) * 8
    buf = hl.compress_multi(u, 1e-3, np.inf, mg.REL, devices=devs)
    meta = hl.metadata_parse(bytes(buf[:8192]))
    assert meta["shape"] == [64, 512, 512, 512] and meta["domain_decomposed"] is True and meta["dd_size"] == 8
    assert np.float32(meta["norm"]) == np.float32(nrm)
    assert buf.size < 0.5 * u.nbytes
    v = hl.decompress_multi(buf, devices=devs)
    err = max(float(np.max(np.abs(v[t] - u[t]))) for t in range(64))
    assert err <= 1e-3 * nrm
    # the single-device reader on the same stream (device-resident out): identical reconstruction
    back = torch.empty((64, 512, 512, 512), dtype=torch.float32, device="cuda")
    hl.decompress(torch.from_numpy(np.ascontiguousarray(buf)).cuda(), out=back)
    for t in (0, 31, 63):
        assert np.array_equal(back[t].cpu().numpy(), v[t])
    del back
    hl.release_cache()
    torch.cuda.empty_cache()


def test_mirror_reference_coord_cast_switch():
    """Config.mirror_reference_coord_cast: stock MGARD-X rebuilds the coordinates of a non-uniform
    grid through `(float)` when it decompresses, even for double data
    (CompressionHighLevel.hpp:455-462). With the switch mgh_decompress reconstructs on exactly that
    grid -- equal to the low-level dequantize + recompose on a hierarchy built from the
    float-rounded coordinates --, without it on the coordinates the compressor used."""
    torch, mg, hl = _mods()
    shape, dt = (33, 40, 65), np.float64
    u = smooth_field(shape, dt)
    coords = nonuniform_coords(shape, dt)
    buf = hl.compress(u, 1e-2, np.inf, mg.ABS, coords=coords)
    assert buf.size < u.nbytes // 2                      # (a Huffman record, not a raw subdomain)
    exact = hl.decompress(buf)
    cast = hl.decompress(buf, config=hl.Config(mirror_reference_coord_cast=1))
    assert not np.array_equal(exact, cast)               # (the grids differ in the last bits)
    assert float(np.max(np.abs(cast - u))) <= 2e-2       # still a faithful reconstruction
    # the same integers recomposed at the low level on the two grids
    h = mg.Hierarchy(shape, dt, coords=coords)
    q, oi, ov, cnt, _ = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.ABS, 1e-2, np.inf, outlier_cap=u.size)
    want_exact = h.dequantize_recompose(q.clone(), mg.ABS, 1e-2, np.inf, 1.0, outlier_idx=oi[:cnt], outlier_val=ov[:cnt])
    h.close()
    hf = mg.Hierarchy(shape, dt, coords=[c.astype(np.float32).astype(np.float64) for c in coords])
    want_cast = hf.dequantize_recompose(q.clone(), mg.ABS, 1e-2, np.inf, 1.0, outlier_idx=oi[:cnt], outlier_val=ov[:cnt])
    hf.close()
    assert np.array_equal(exact, want_exact.cpu().numpy())
    assert np.array_equal(cast, want_cast.cpu().numpy())


def _canonical_records(hl, buf):
    """The container as (header bytes, [record with its outlier list sorted by index]): what must not
    depend on how the subdomains were scheduled (the outlier list of a record is in atomic order)."""
    b = bytes(buf)
    m = hl.metadata_parse(b)
    out = []
    for r in pl.split_container(b, m["metadata_size"]):
        if m["lossless"] == hl.HUFFMAN_ZSTD and len(r) >= 8:
            # [u64 size of the Huffman record][zstd frame] (Zstd.hpp:69-90)
            import ctypes
            z = ctypes.CDLL("libzstd.so.1")
            z.ZSTD_decompress.restype = ctypes.c_size_t
            z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            raw_size = int(np.frombuffer(r[:8], dtype="<u8")[0])
            if raw_size < (1 << 32):
                dst = ctypes.create_string_buffer(max(raw_size, 1))
                got = z.ZSTD_decompress(dst, raw_size, r[8:], len(r) - 8)
                if got == raw_size:
                    r = dst.raw[:raw_size]
        try:
            p = pl.parse_huffman_record(r)
        except Exception:  # (a raw subdomain: GPUPipelines.hpp:136-155)
            out.append(("raw", r))
            continue
        n = len(p["outlier_idx"])
        tail = 0 if p["sync"] is None else 8 + 4 * p["sync"].size   # (behind the lists: synchronisation points)
        head = r[:len(r) - 16 * n - tail] + r[len(r) - tail:]
        order = np.argsort(p["outlier_idx"], kind="stable")
        out.append(("huffman", head, p["outlier_idx"][order].tobytes(), p["outliers"][order].tobytes()))
    return b[:m["metadata_size"]], out


from struct import error as struct_error  # noqa: E402


@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("kind,dt,s", [("block", np.float32, np.inf), ("variable", np.float32, 0.0),
                                       ("maxdim_auto", np.float64, np.inf), ("variable_nonuniform", np.float32, np.inf),
                                       ("block_zstd", np.float32, np.inf), ("variable_dim0", np.float32, np.inf)])
def test_pipelined_and_sequential_schedules_write_the_same_container(kind, dt, s, where, monkeypatch):
    """The two-lane subdomain pipeline (subdomain k + 1 decomposed while k is encoded;
    GPUPipelines.hpp:88-207) against MGH_HL_PIPELINE=0, every subdomain start to end before the next
    is queued: the same container byte for byte (outlier lists of a record compared sorted), and each
    of the two schedules of mgh_decompress reconstructs the same values from it."""
    torch, mg, hl = _mods()
    shape = (66, 120, 80)
    u = smooth_field(shape, dt, noise=1e-3)
    coords = None
    if kind == "maxdim_auto":
        cfg = hl.Config(max_memory_footprint=40 * u.size)
    elif kind in ("block", "block_zstd"):
        cfg = hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=40,
                        lossless=hl.HUFFMAN_ZSTD if kind == "block_zstd" else hl.HUFFMAN)
    elif kind == "variable_dim0":  # contiguous slabs: compressed / reconstructed in place
        cfg = hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=0,
                        domain_decomposition_sizes=[10, 20, 9, 27])
    else:
        cfg = hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=1,
                        domain_decomposition_sizes=[40, 30, 50])
        if kind == "variable_nonuniform":
            from tests.util import nonuniform_coords
            coords = nonuniform_coords(shape, dt)
    src = torch.from_numpy(u).cuda() if where == "device" else u
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MGH_HL_PIPELINE", mode)
        buf = hl.compress(src, 1e-3, s, mg.REL, config=cfg, coords=coords)
        host = buf.cpu().numpy() if hasattr(buf, "cpu") else np.asarray(buf)
        got[mode] = (_canonical_records(hl, host), buf)
    assert got["1"][0][0] == got["0"][0][0]
    assert len(got["1"][0][1]) == len(got["0"][0][1]) > 1
    for a, b in zip(got["1"][0][1], got["0"][0][1]):
        assert a == b
    back = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MGH_HL_PIPELINE", mode)
        v = hl.decompress(got["1"][1], config=cfg)
        back[mode] = v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)
    assert np.array_equal(back["1"], back["0"])
    nrm = _norm(u, s)
    assert _err(u, back["1"], s, shape) <= 1e-3 * nrm * (1 + 1e-6)


@pytest.mark.parametrize("shift", [0, 3, 4])
def test_records_in_device_memory_at_any_byte_offset(shift):
    """Behind a header of arbitrary length the records of a device-resident container start at any
    byte offset. The encoder writes its code units straight into such a record and the ring
    decoder reads them where they are (unaligned 8-byte accesses); a chunk of more than 16 bits per
    symbol -- the encoder's path with atomics on the destination -- makes the host run the encoder
    again into an aligned buffer. Same record as through the host path, byte for byte."""
    torch, mg, hl = _mods()
    fib = [1, 1]
    while len(fib) < 31:
        fib.append(fib[-1] + fib[-2])
    # rarest symbols first: the leading chunks consist of the longest codes (~20 bits / symbol)
    ordered = np.concatenate([np.full(f, 100 + k, np.int64) for k, f in enumerate(fib)])
    mixed = np.random.default_rng(5).permutation(ordered)
    ctx = hl.Lossless()
    for q, chunk in ((ordered, 2048), (mixed, 2048), (_symbols(700_001, seed=9), 20480)):
        d_q = torch.from_numpy(q).cuda()
        oi = torch.tensor([5, 77, 12345], dtype=torch.int64, device="cuda")
        ov = torch.tensor([-9, 1 << 40, 3], dtype=torch.int64, device="cuda")
        want = ctx.compress(d_q, 8192, chunk, outlier_idx=oi, outlier_val=ov)
        pool = torch.zeros(len(want) + 4096, dtype=torch.uint8, device="cuda")
        base = (-pool.data_ptr()) % 256 + shift   # 256-byte aligned + shift
        rec = ctx.compress_device(d_q, pool[base:], 8192, chunk, outlier_idx=oi, outlier_val=ov)
        assert rec.data_ptr() % 8 == shift % 8
        assert bytes(rec.cpu().numpy()) == want
        assert not pool[:base].any() and not pool[base + rec.numel():].any()
        back, bi, bv = ctx.decompress(rec, q.size)
        assert np.array_equal(back.cpu().numpy(), q)
        assert torch.equal(bi, oi) and torch.equal(bv, ov)
    if shift:
        r = pl.parse_huffman_record(want)
        assert r  # (parsed: the layout is the reference's)
    with pytest.raises(mg.MgardHipError):
        ctx.compress_device(d_q, torch.zeros(1000, dtype=torch.uint8, device="cuda"), 8192, 20480)
    ctx.close()


@pytest.mark.parametrize("lossless", ["HUFFMAN_LZ4", "CPU_LOSSLESS"])
def test_unsupported_lossless_types_are_refused_with_a_fixed_status(lossless):
    """lossless_type::Huffman_LZ4 (needs nvcomp upstream, Lossless/LZ4.hpp) and CPU_Lossless (the
    legacy MGARD-CPU Huffman + zstd stream, Lossless/CPU.hpp) are not built here.
    What a caller gets is pinned: the writer refuses the config with MGH_ERR_INVALID_ARGUMENT (-1) and
    a message that names the two supported types, the reader refuses a stream whose header carries
    such a type with MGH_ERR_FORMAT (-8) -- never a crash, never a silently different stream."""
    torch, mg, hl = _mods()
    kind = getattr(hl, lossless)
    u = smooth_field((33, 40, 65), np.float32)
    with pytest.raises(mg.MgardHipError, match=r"-1.*only Huffman and Huffman_Zstd"):
        hl.compress(u, 1e-3, np.inf, mg.REL, config=hl.Config(lossless=kind))
    with pytest.raises(mg.MgardHipError, match=r"-1.*only Huffman and Huffman_Zstd"):
        hl.compress(torch.from_numpy(u).cuda(), 1e-3, np.inf, mg.REL, config=hl.Config(lossless=kind))
    # a stock stream of that type: its header in front of some payload bytes
    good = hl.compress(u, 1e-3, np.inf, mg.REL)
    m = hl.metadata_parse(bytes(good))
    head = hl.metadata_serialize(mg.FLOAT, list(u.shape), mg.REL, 1e-3, float("inf"), norm=m["norm"], lossless=kind)
    assert hl.metadata_parse(head)["lossless"] == kind
    stream = np.frombuffer(head + bytes(good)[m["metadata_size"]:], dtype=np.uint8).copy()
    with pytest.raises(mg.MgardHipError, match=r"-8.*not supported"):
        hl.decompress(stream)
    with pytest.raises(mg.MgardHipError, match=r"-8.*not supported"):
        hl.decompress(torch.from_numpy(stream).cuda())
    # the library is fine afterwards
    assert np.array_equal(hl.decompress(good), hl.decompress(hl.compress(u, 1e-3, np.inf, mg.REL)))


@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_outlier_estimate_too_small_is_retried_inside_the_subdomain_pipeline(pipeline, monkeypatch):
    """The same re-launch (LinearQuantization.hpp:621-676) when the domain is decomposed: a subdomain
    whose outliers do not fit its lane's lists is quantized again on that lane while the next
    subdomain is already queued on the other one; every lane grows its own lists. The stream equals
    the one written with generously sized lists."""
    torch, mg, hl = _mods()
    monkeypatch.setenv("MGH_HL_PIPELINE", pipeline)
    rng = np.random.default_rng(7)
    shape = (48, 65, 66)
    u = smooth_field(shape, np.float32)
    u[8:40] += (rng.normal(0, 30.0, (32, 65, 66)).astype(np.float32) * (rng.random((32, 65, 66)) < 0.02))
    dd = dict(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=0, domain_decomposition_sizes=[8, 16, 16, 8])
    small = hl.Config(estimate_outlier_ratio=1e-5, huff_dict_size=64, **dd)
    big = hl.Config(estimate_outlier_ratio=1.0, huff_dict_size=64, **dd)
    for src in (u, torch.from_numpy(u).cuda()):
        sa = hl.compress(src, 1e-4, np.inf, mg.REL, config=small)
        sb = hl.compress(src, 1e-4, np.inf, mg.REL, config=big)
        ha = sa.cpu().numpy() if hasattr(sa, "cpu") else np.asarray(sa)
        hb = sb.cpu().numpy() if hasattr(sb, "cpu") else np.asarray(sb)
        assert _canonical_records(hl, ha) == _canonical_records(hl, hb)
        a = hl.decompress(sa, config=small)
        a = a if isinstance(a, np.ndarray) else a.cpu().numpy()
        assert float(np.max(np.abs(a - u))) <= 1e-4 * float(np.max(np.abs(u))) * (1 + 1e-6)


def test_more_subdomain_shapes_than_the_hierarchy_cache_holds():
    """A 4-D Block decomposition with a remainder in every dimension has 16 subdomain shapes, times two
    lanes: more hierarchies than the per-thread cache keeps (8 per lane). The ones that do not fit are
    built for their subdomain alone and destroyed behind it; nothing cached is evicted while the other
    lane may still be running on it. Round trip within the bound, twice in a row (the second call
    finds a full cache)."""
    torch, mg, hl = _mods()
    shape = (21, 21, 21, 21)
    u = smooth_field(shape, np.float32)
    cfg = hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=9)
    nrm = float(np.max(np.abs(u)))
    for _ in range(2):
        buf = hl.compress(torch.from_numpy(u).cuda(), 1e-3, np.inf, mg.REL, config=cfg)
        m = hl.metadata_parse(bytes(buf[:8192].cpu().numpy()))
        assert m["domain_decomposed"] and m["dd_method"] == hl.DD_BLOCK
        v = hl.decompress(buf, config=cfg).cpu().numpy()
        assert float(np.max(np.abs(v - u))) <= 1e-3 * nrm * (1 + 1e-6)
    hl.release_cache()


@pytest.mark.parametrize("kind,n,chunk", [("normal60", 3 * 20480 + 777, 20480), ("uniform4096", 50001, 4096),
                                          ("sorted", 2 * 20480 + 5, 20480), ("narrow", 3 * 20480, 20480),
                                          ("normal60", 20480 * 2, 1024)])
def test_synchronisation_points_behind_the_huffman_record(kind, n, chunk, monkeypatch):
    """Behind the reference's payload the encoder leaves, for streams of 4 bits per symbol or more, 64
    synchronisation points per chunk: where the first code at or behind bit k * ceil(bits / 64) starts and
    which symbol it is (huffman.hpp: k_encode_chain). The independent reader recomputes them from the
    symbols and the decodebook; the library's decoder must return the same symbols with them (one pass)
    and without them (MGH_HUFF_SYNC_DECODE=0: speculative subsequences + counting pass), also on streams
    that do not re-synchronise by themselves (sorted symbols: long runs of one code length); a record
    written with MGH_HUFF_SYNC=0 is the same record without the section."""
    torch, mg, hl = _mods()
    rng = np.random.default_rng(11)
    if kind == "normal60":
        q = _symbols(n, width=60.0)
    elif kind == "uniform4096":
        q = rng.integers(2048, 2048 + 4096, n).astype(np.int64)
    elif kind == "sorted":
        q = np.sort(_symbols(n, width=300.0))
    else:
        q = _symbols(n, width=0.4)  # ~1.5 bits per symbol: no section
    oi = np.array([1, n // 3], dtype=np.int64)
    ov = np.array([-5, 99999], dtype=np.int64)
    q[oi] = 0
    ctx = hl.Lossless()
    qd = torch.from_numpy(q).cuda()
    args = (8192, chunk, hl.HUFFMAN, 3, torch.from_numpy(oi).cuda(), torch.from_numpy(ov).cuda())
    monkeypatch.setenv("MGH_HUFF_SYNC", "1")
    rec = ctx.compress(qd, *args)
    r = pl.parse_huffman_record(rec)
    if kind == "narrow":
        assert r["sync"] is None
    else:
        assert r["sync"] is not None
        np.testing.assert_array_equal(r["sync"], pl.expected_sync_points(r, q))
    monkeypatch.setenv("MGH_HUFF_SYNC", "0")
    plain = ctx.compress(qd, *args)
    assert pl.parse_huffman_record(plain)["sync"] is None
    assert rec[:len(plain)] == plain
    # (MGH_HUFF_PAIR: records with the section go through k_decode_sync -- two codes per table slot
    # where they fit: 2 always, 1 for short codes only -- or, 0, through k_decode_ring's single-symbol steps)
    # MGH_HUFF_LEAN=1: the writing pass without divergent control flow (k_decode_lean) instead of k_decode_ring's
    for sync_decode, pair, lean in (("1", "2", "0"), ("1", "0", "1"), ("1", "0", "0"), ("1", "1", "0"), ("0", "1", "0")):
        monkeypatch.setenv("MGH_HUFF_SYNC_DECODE", sync_decode)
        monkeypatch.setenv("MGH_HUFF_PAIR", pair)
        monkeypatch.setenv("MGH_HUFF_LEAN", lean)
        for payload in (rec, plain, torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda(),
                        torch.frombuffer(bytearray(b"xyz" + rec), dtype=torch.uint8).cuda()[3:]):
            back, bi, bv = ctx.decompress(payload, n, hl.HUFFMAN)
            assert np.array_equal(back.cpu().numpy(), q), (kind, sync_decode, pair, lean)
            assert np.array_equal(bi.cpu().numpy(), oi) and np.array_equal(bv.cpu().numpy(), ov)
    ctx.close()


@pytest.mark.parametrize("where", ["device", "host"])
def test_damaged_sync_tag_is_an_error_not_a_quiet_decode(where):
    """The synchronisation-point section behind a Huffman record is found by the record's size; its
    tag "MGHSYNC1" decides whether it IS one. A record whose tag is damaged must not be decoded with
    arbitrary synchronisation points: a host record falls back to decoding without them (the lists
    then fail their own check or decode correctly), a device-resident record -- whose tag is looked
    at by a kernel -- returns MGH_ERR_FORMAT."""
    torch, mg, hl = _mods()
    u = smooth_field((100, 128, 128), np.float32, noise=3e-3)   # (4+ bits per symbol: the encoder's rule for the section)
    c = hl.compress(torch.from_numpy(u).cuda(), 1e-3, np.inf, mg.REL).cpu().numpy()
    at = bytes(c).rfind(b"MGHSYNC1")
    assert at > 0, "the record carries no synchronisation points (MGH_HUFF_SYNC=0?)"
    bad = c.copy()
    bad[at + 3] ^= 0x40
    if where == "device":
        with pytest.raises(mg.MgardHipError):
            hl.decompress(torch.from_numpy(bad).cuda())
        # ... and the context is fine afterwards
        v = hl.decompress(torch.from_numpy(c).cuda()).cpu().numpy()
    else:
        try:
            v = hl.decompress(bad)
        except mg.MgardHipError:
            v = hl.decompress(c)
    assert float(np.max(np.abs(v - u))) <= 1e-3 * float(np.max(np.abs(u)))
