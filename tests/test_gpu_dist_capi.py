"""mgh_compress_dist / mgh_decompress_dist (one rank per GPU over RCCL; reference pattern:
examples/mgard-x/CompressXgcData/TestXGCAbsoluteError.cpp:36-252) on ONE GPU: a real RCCL
communicator of one rank (ncclCommInitRank), created with the RCCL already in the process; the calls
run their collectives on it and must then write / read exactly what mgh_compress / mgh_decompress do.
More than one rank needs more than one GPU (RCCL refuses two ranks on one device): the N > 1 control
flow is unmeasured here; the Python path is covered on gloo in
tests/test_distributed_cpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import smooth_field

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rccl():
    """The RCCL torch brought into the process (a second copy of the library beside it would share
    its exported symbols with the first)."""
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        path = "/opt/rocm/lib/librccl.so.1"
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib, path


@pytest.fixture(scope="module")
def comm():
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    lib, path = _rccl()
    uid = _UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    c = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(c), 1, uid, 0) == 0
    yield c.value, path
    lib.ncclCommDestroy(c)


@pytest.mark.parametrize("shape,dt,s,mode", [((40, 50, 60), np.float32, np.inf, "REL"),
                                             ((33, 65), np.float64, 0.0, "REL"),
                                             ((9, 20, 30, 40), np.float32, np.inf, "ABS")])
def test_one_rank_communicator_writes_what_mgh_compress_writes(comm, shape, dt, s, mode):
    import torch
    import mgard_amd as mg
    from mgard_amd import highlevel as hl
    from tests.test_gpu_host_path import _same_container
    handle, path = comm
    u = smooth_field(shape, dt)
    d = torch.from_numpy(u).cuda()
    m = getattr(mg, mode)
    tol = 1e-3
    got = hl.compress_dist(handle, 0, 1, d, tol, s, m, rccl_path=path)
    want = hl.compress(d, tol, s, m)
    back = hl.decompress_dist(handle, 0, 1, got, shape, d.dtype)
    if s == np.inf or mode == "ABS":
        assert _same_container(hl, got.cpu().numpy(), want.cpu().numpy())
        assert torch.equal(back, hl.decompress(want))
    else:
        # (an L2 norm is a sum whose last bits depend on the order of the atomics: two compressions of
        # the same array may differ in the header's norm -- compare what the bound promises)
        assert abs(hl.metadata_parse(bytes(got.cpu().numpy()))["norm"] - hl.metadata_parse(bytes(want.cpu().numpy()))["norm"]) \
            <= 1e-12 * hl.metadata_parse(bytes(want.cpu().numpy()))["norm"]
        l2 = float(np.sqrt(np.mean((back.cpu().numpy().astype(np.float64) - u) ** 2)))
        assert l2 <= tol * float(np.sqrt(np.mean(u.astype(np.float64) ** 2)))


def test_dist_argument_checks(comm):
    import torch
    import mgard_amd as mg
    from mgard_amd import highlevel as hl
    handle, path = comm
    d = torch.zeros((10, 10, 10), device="cuda")
    with pytest.raises(mg.MgardHipError):
        hl.compress_dist(handle, 1, 1, d, 1e-3)            # rank outside the communicator
    with pytest.raises(mg.MgardHipError):
        hl.compress_dist(0, 0, 1, d, 1e-3)                 # no communicator
    L = hl._hl()
    assert L.mgh_dist_use_library(b"/nonexistent/librccl.so") < 0
    assert L.mgh_dist_use_library(path.encode()) == 0


@pytest.mark.parametrize("nranks", [2, 4])
def test_ranks_on_several_gpus_of_one_node(nranks):
    """N > 1 (opt-in, and only where the box has the GPUs): one thread per rank and
    device, ncclCommInitRank inside the threads, the container of mgh_compress_dist against
    mgh_compress with the same MaxDim decomposition on one device, slabs back through
    mgh_decompress_dist."""
    import threading
    import torch
    import mgard_amd as mg
    from mgard_amd import highlevel as hl
    from tests.test_gpu_host_path import _same_container
    if os.environ.get("MGARD_HIP_TEST_MULTI_GPU", "0") != "1":
        pytest.skip("opt-in (MGARD_HIP_TEST_MULTI_GPU=1): has never run -- no box with more than one GPU has been "
                    "available to this repository; bench.py --gpus N runs the same calls in a child of its own")
    if torch.cuda.device_count() < nranks:
        pytest.skip("needs %d GPUs" % nranks)
    lib, path = _rccl()
    uid = _UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    slab = 12
    shape = (slab * nranks - 4, 40, 50)     # (the last rank holds fewer planes)
    u = smooth_field(shape, np.float32)
    results, errors = {}, []

    def work(r):
        try:
            torch.cuda.set_device(r)
            c = C.c_void_p()
            assert lib.ncclCommInitRank(C.byref(c), nranks, uid, r) == 0
            lo, hi = r * slab, min((r + 1) * slab, shape[0])
            d = torch.from_numpy(u[lo:hi]).to("cuda:%d" % r)
            cfg = hl.Config(dev_id=r)
            got = hl.compress_dist(c.value, r, nranks, d, 1e-3, np.inf, mg.REL, config=cfg, rccl_path=path)
            back = hl.decompress_dist(c.value, r, nranks, got, d.shape, d.dtype, config=cfg, device=d.device)
            results[r] = (None if got is None else got.cpu().numpy(), back.cpu().numpy())
            lib.ncclCommDestroy(c)
            hl.release_cache()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    torch.cuda.set_device(0)
    want = hl.compress(torch.from_numpy(u).cuda(), 1e-3, np.inf, mg.REL,
                       config=hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=0,
                                        domain_decomposition_sizes=[min(slab, shape[0] - r * slab) for r in range(nranks)]))
    # (the header says MaxDim there and Variable here: compare the records and the reconstruction)
    from tests import payload as pl
    a, b = bytes(results[0][0]), bytes(want.cpu().numpy())
    ma, mb = hl.metadata_parse(a), hl.metadata_parse(b)
    assert ma["norm"] == mb["norm"] and ma["shape"] == mb["shape"]
    ra, rb = pl.split_container(a, ma["metadata_size"]), pl.split_container(b, mb["metadata_size"])
    assert len(ra) == len(rb) == nranks
    whole = hl.decompress(torch.from_numpy(results[0][0]).cuda()).cpu().numpy()
    for r in range(nranks):
        lo, hi = r * slab, min((r + 1) * slab, shape[0])
        assert np.array_equal(results[r][1], whole[lo:hi])
    assert float(np.max(np.abs(whole - u))) <= 1e-3 * float(np.max(np.abs(u)))
