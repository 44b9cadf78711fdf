"""CPU tests of the one-rank-per-GPU C entry points (include/mgard_hip_compress.h: mgh_compress_dist,
mgh_decompress_dist, mgh_dist_use_library): what they must refuse BEFORE any collective or device call
-- a rank that fails its argument checks returns at once instead of leaving its peers inside RCCL --
and that RCCL is resolved with dlopen (the library itself does not link it)."""
import ctypes as C
import os
import subprocess

import pytest


def _lib():
    import mgard_amd
    from mgard_amd import highlevel
    return mgard_amd, highlevel, highlevel._hl()


def test_library_does_not_link_rccl():
    import mgard_amd
    out = subprocess.run(["readelf", "-d", mgard_amd.lib_path()], capture_output=True, text=True).stdout
    needed = [l for l in out.splitlines() if "NEEDED" in l]
    assert needed and not any("rccl" in l or "nccl" in l for l in needed), needed


def test_use_library_refuses_what_is_not_rccl(tmp_path):
    mg, hl, L = _lib()
    assert L.mgh_dist_use_library(b"/nonexistent/librccl.so") < 0
    assert L.mgh_dist_use_library(None) < 0
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if os.path.exists(libm):   # a library without the nccl* symbols
        assert L.mgh_dist_use_library(libm.encode()) < 0


@pytest.mark.parametrize("rank,nranks,root", [(1, 1, 0), (-1, 2, 0), (0, 0, 0), (0, 2, 2), (0, 2, -1)])
def test_dist_calls_check_ranks_before_anything_else(rank, nranks, root):
    mg, hl, L = _lib()
    cfg = hl.Config()
    shp = (C.c_uint64 * 3)(8, 8, 8)
    fake_comm, fake_data = C.c_void_p(1), C.c_void_p(16)
    out, size = C.c_void_p(), C.c_size_t(0)
    rc = L.mgh_compress_dist(fake_comm, rank, nranks, root, 3, mg.FLOAT, shp, 1e-3, float("inf"), mg.REL, fake_data,
                             C.byref(out), C.byref(size), None, C.byref(cfg), 0)
    assert rc == -1, rc   # MGH_ERR_INVALID_ARGUMENT
    rc = L.mgh_decompress_dist(fake_comm, rank, nranks, root, fake_data, 100, fake_data, C.byref(cfg))
    assert rc == -1, rc


def test_dist_calls_refuse_null_arguments():
    mg, hl, L = _lib()
    cfg = hl.Config()
    shp = (C.c_uint64 * 3)(8, 8, 8)
    out, size = C.c_void_p(), C.c_size_t(0)
    assert L.mgh_compress_dist(None, 0, 1, 0, 3, mg.FLOAT, shp, 1e-3, 0.0, mg.REL, C.c_void_p(16), C.byref(out),
                               C.byref(size), None, C.byref(cfg), 0) == -1
    assert L.mgh_compress_dist(C.c_void_p(1), 0, 1, 0, 3, mg.FLOAT, shp, 1e-3, 0.0, mg.REL, None, C.byref(out),
                               C.byref(size), None, C.byref(cfg), 0) == -1
    assert L.mgh_decompress_dist(C.c_void_p(1), 0, 1, 0, C.c_void_p(16), 100, None, C.byref(cfg)) == -1
