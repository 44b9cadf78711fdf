"""Python host side of the high-level path (include/mgard_hip_compress.h): whole-array
compress / decompress, the container header and the lossless stage, all through the C ABI.

    buf = mgard_amd.highlevel.compress(u, tol=1e-3, s=inf, mode=REL)     # numpy or cuda tensor
    v = mgard_amd.highlevel.decompress(buf)

mirrors mgard_x::compress / decompress (reference include/compress_x.hpp:31-154).
"""
import ctypes as C

import numpy as np

from . import ABS, DOUBLE, FLOAT, INF, REL, MgardHipError, _check, load_library  # noqa: F401

MAX_DIM = 5
DD_MAXDIM, DD_BLOCK, DD_VARIABLE = 0, 1, 2
HUFFMAN, HUFFMAN_LZ4, HUFFMAN_ZSTD, CPU_LOSSLESS = 0, 1, 2, 3

HL_SYMBOLS = [
    "mgh_config_default", "mgh_compress", "mgh_decompress", "mgh_infer_shape",
    "mgh_infer_data_type", "mgh_free_device", "mgh_release_cache", "mgh_metadata_serialize",
    "mgh_metadata_parse", "mgh_lossless_create", "mgh_lossless_destroy", "mgh_lossless_compress",
    "mgh_lossless_decompress", "mgh_lossless_compress_device", "mgh_memcpy", "mgh_huffman_codebook",
    "mgh_compress_multi", "mgh_decompress_multi", "mgh_pin_memory", "mgh_check_memory_pinned",
    "mgh_unpin_memory", "mgh_dist_use_library", "mgh_compress_dist", "mgh_decompress_dist",
    "mgh_decompress_into",
]


class Config(C.Structure):
    """mgh_config (subset of mgard_x::Config)."""
    _fields_ = [
        ("dev_id", C.c_int),
        ("domain_decomposition", C.c_int),
        ("domain_decomposition_dim", C.c_int),
        ("domain_decomposition_sizes", C.POINTER(C.c_uint64)),
        ("num_domain_decomposition_sizes", C.c_uint64),
        ("block_size", C.c_uint64),
        ("estimate_outlier_ratio", C.c_double),
        ("huff_dict_size", C.c_uint64),
        ("huff_block_size", C.c_uint64),
        ("lossless", C.c_int),
        ("zstd_compress_level", C.c_int),
        ("normalize_coordinates", C.c_int),
        ("max_larget_level", C.c_uint64),
        ("max_memory_footprint", C.c_uint64),
        ("auto_pin_host_buffers", C.c_int),
        ("reorder", C.c_int),
        ("mirror_reference_coord_cast", C.c_int),
    ]

    def __init__(self, **kw):
        super().__init__()
        _hl().mgh_config_default(C.byref(self))
        self._sizes = None
        for k, v in kw.items():
            if k == "domain_decomposition_sizes":
                self._sizes = (C.c_uint64 * len(v))(*v)
                self.domain_decomposition_sizes = C.cast(self._sizes, C.POINTER(C.c_uint64))
                self.num_domain_decomposition_sizes = len(v)
            else:
                setattr(self, k, v)


class HeaderInfo(C.Structure):
    """mgh_header_info."""
    _fields_ = [
        ("version", C.c_uint64 * 3),
        ("dtype", C.c_int),
        ("D", C.c_int),
        ("shape", C.c_uint64 * MAX_DIM),
        ("uniform", C.c_int),
        ("coords", C.POINTER(C.c_double) * MAX_DIM),
        ("error_bound_type", C.c_int),
        ("tol", C.c_double),
        ("s", C.c_double),
        ("norm", C.c_double),
        ("domain_decomposed", C.c_int),
        ("dd_method", C.c_int),
        ("dd_dim", C.c_uint64),
        ("dd_size", C.c_uint64),
        ("l_target", C.c_uint64),
        ("reorder", C.c_int),
        ("lossless", C.c_int),
        ("huff_dict_size", C.c_uint64),
        ("huff_block_size", C.c_uint64),
    ]


_declared = False


def _hl():
    global _declared
    L = load_library()
    if _declared:
        return L
    vp, u64 = C.c_void_p, C.c_uint64
    L.mgh_config_default.argtypes = [C.POINTER(Config)]
    L.mgh_config_default.restype = None
    L.mgh_compress.argtypes = [C.c_int, C.c_int, C.POINTER(u64), C.c_double, C.c_double, C.c_int, vp,
                               C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                               C.POINTER(Config), C.c_int]
    L.mgh_decompress.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(Config), C.c_int]
    L.mgh_decompress_into.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp]
    L.mgh_dist_use_library.argtypes = [C.c_char_p]
    L.mgh_compress_dist.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u64), C.c_double,
                                    C.c_double, C.c_int, vp, C.POINTER(vp), C.POINTER(C.c_size_t), vp, vp, C.c_int]
    L.mgh_decompress_dist.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp, vp]
    L.mgh_pin_memory.argtypes = [vp, C.c_size_t]
    L.mgh_check_memory_pinned.argtypes = [vp]
    L.mgh_unpin_memory.argtypes = [vp]
    L.mgh_compress_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(u64), C.c_double,
                                     C.c_double, C.c_int, vp, C.POINTER(vp), C.POINTER(C.c_size_t),
                                     C.POINTER(vp), C.POINTER(Config), C.c_int]
    L.mgh_decompress_multi.argtypes = [C.c_int, C.POINTER(C.c_int), vp, C.c_size_t, C.POINTER(vp),
                                       C.POINTER(Config), C.c_int]
    L.mgh_infer_shape.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(u64)]
    L.mgh_infer_data_type.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int)]
    L.mgh_free_device.argtypes = [vp]
    L.mgh_free_device.restype = None
    L.mgh_release_cache.restype = None
    L.mgh_metadata_serialize.argtypes = [C.POINTER(HeaderInfo), vp, u64]
    L.mgh_metadata_serialize.restype = C.c_int64
    L.mgh_metadata_parse.argtypes = [vp, u64, C.POINTER(HeaderInfo), C.POINTER(C.c_double), u64,
                                     C.POINTER(u64)]
    L.mgh_lossless_create.argtypes = [C.POINTER(vp), C.c_int]
    L.mgh_lossless_destroy.argtypes = [vp]
    L.mgh_lossless_destroy.restype = None
    L.mgh_lossless_compress.argtypes = [vp, vp, u64, u64, u64, C.c_int, C.c_int, vp, vp, u64,
                                        C.POINTER(vp), C.POINTER(u64), vp]
    L.mgh_lossless_compress_device.argtypes = [vp, vp, u64, u64, u64, vp, vp, u64, vp, u64, C.POINTER(u64), vp]
    L.mgh_lossless_decompress.argtypes = [vp, vp, u64, C.c_int, vp, u64, C.POINTER(vp), C.POINTER(vp),
                                          C.POINTER(u64), vp]
    L.mgh_memcpy.argtypes = [vp, vp, C.c_size_t]
    L.mgh_huffman_codebook.argtypes = [vp, u64, vp, vp, vp, vp]
    _declared = True
    return L


# ---- header ---------------------------------------------------------------------------------
def metadata_serialize(dtype, shape, mode, tol, s, norm=0.0, coords=None, dd=None, lossless=HUFFMAN,
                       dict_size=8192, block_size=20480, reorder=0, l_target=0):
    """Bytes of preamble + header for the given description. dd = (method, dim, size) or None."""
    info = HeaderInfo()
    info.version[0], info.version[1], info.version[2] = 1, 0, 0
    info.dtype = dtype
    info.D = len(shape)
    for d, n in enumerate(shape):
        info.shape[d] = n
    keep = []
    info.uniform = 1 if coords is None else 0
    if coords is not None:
        for d, c in enumerate(coords):
            a = np.ascontiguousarray(c, dtype=np.float64)
            keep.append(a)
            info.coords[d] = a.ctypes.data_as(C.POINTER(C.c_double))
    info.error_bound_type = mode
    info.tol, info.s, info.norm = tol, s, norm
    if dd is None:
        info.domain_decomposed, info.dd_dim, info.dd_size = 0, 0, shape[0]
    else:
        info.domain_decomposed, info.dd_method, info.dd_dim, info.dd_size = 1, dd[0], dd[1], dd[2]
    info.l_target = l_target
    info.reorder = reorder
    info.lossless = lossless
    info.huff_dict_size, info.huff_block_size = dict_size, block_size
    L = _hl()
    n = _check(L.mgh_metadata_serialize(C.byref(info), None, 0))
    buf = (C.c_uint8 * n)()
    _check(L.mgh_metadata_serialize(C.byref(info), buf, n))
    return bytes(buf)


def metadata_parse(data):
    """dict of the header fields + 'metadata_size'."""
    L = _hl()
    raw = (C.c_uint8 * len(data)).from_buffer_copy(data)
    info = HeaderInfo()
    cap = 1 << 16
    while True:
        store = (C.c_double * cap)()
        ms = C.c_uint64()
        rc = L.mgh_metadata_parse(raw, len(data), C.byref(info), store, cap, C.byref(ms))
        if rc < 0 and b"coords_storage" in L.mgh_last_error() and cap < (1 << 28):
            cap *= 16
            continue
        _check(rc)
        break
    D = info.D
    out = dict(dtype=info.dtype, shape=[int(info.shape[d]) for d in range(D)], uniform=bool(info.uniform),
               mode=info.error_bound_type, tol=info.tol, s=info.s, norm=info.norm,
               domain_decomposed=bool(info.domain_decomposed), dd_method=info.dd_method,
               dd_dim=int(info.dd_dim), dd_size=int(info.dd_size), l_target=int(info.l_target),
               reorder=info.reorder, lossless=info.lossless, dict_size=int(info.huff_dict_size),
               block_size=int(info.huff_block_size), version=[int(v) for v in info.version],
               metadata_size=int(ms.value))
    if not info.uniform:
        out["coords"] = [np.array([info.coords[d][i] for i in range(out["shape"][d])]) for d in range(D)]
    return out


# ---- whole-array compress / decompress -------------------------------------------------------
def _as_ptr(a):
    """(pointer, dtype code, shape, keepalive) of a numpy array or a cuda tensor."""
    import torch
    if isinstance(a, torch.Tensor):
        if not a.is_contiguous():
            a = a.contiguous()
        dt = FLOAT if a.dtype == torch.float32 else DOUBLE if a.dtype == torch.float64 else None
        if dt is None:
            raise MgardHipError("float32 or float64 data expected")
        return C.c_void_p(a.data_ptr()), dt, tuple(a.shape), a
    a = np.ascontiguousarray(a)
    dt = FLOAT if a.dtype == np.float32 else DOUBLE if a.dtype == np.float64 else None
    if dt is None:
        raise MgardHipError("float32 or float64 data expected")
    return C.c_void_p(a.ctypes.data), dt, a.shape, a


def compress(data, tol, s=INF, mode=REL, coords=None, config=None, out_capacity=None, out=None):
    """mgard_x::compress. `data`: numpy array (host) or cuda tensor (device). Returns the
    compressed stream as a numpy uint8 array (host input) or a cuda uint8 tensor (device input).
    `out`: optional pre-allocated uint8 buffer of the same kind to write into (its size is the
    capacity)."""
    import torch
    L = _hl()
    cfg = config if config is not None else Config()
    ptr, dt, shape, keep = _as_ptr(data)
    D = len(shape)
    shp = (C.c_uint64 * D)(*shape)
    cptr = None
    ckeep = []
    if coords is not None:
        npdt = np.float32 if dt == FLOAT else np.float64
        arr = (C.c_void_p * D)()
        for d in range(D):
            c = np.ascontiguousarray(coords[d], dtype=npdt)
            ckeep.append(c)
            arr[d] = c.ctypes.data
        cptr = arr
    on_device = isinstance(data, torch.Tensor) and data.is_cuda
    nbytes = int(np.prod(shape)) * (4 if dt == FLOAT else 8)
    cap = int(out_capacity) if out_capacity is not None else nbytes + 1000000
    if out is not None:
        cap = int(out.numel()) if on_device else int(out.size)
        optr = C.c_void_p(out.data_ptr()) if on_device else C.c_void_p(out.ctypes.data)
    elif on_device:
        out = torch.empty(cap, dtype=torch.uint8, device=data.device)
        optr = C.c_void_p(out.data_ptr())
    else:
        out = np.empty(cap, dtype=np.uint8)
        optr = C.c_void_p(out.ctypes.data)
    size = C.c_size_t(cap)
    _check(L.mgh_compress(D, dt, shp, float(tol), float(s), int(mode), ptr, C.byref(optr), C.byref(size),
                          cptr, C.byref(cfg), 1))
    return out[:size.value]


def infer(buf):
    """(shape, dtype code) of a compressed stream (numpy uint8 array or cuda uint8 tensor)."""
    import torch
    L = _hl()
    if isinstance(buf, torch.Tensor):
        p, n = C.c_void_p(buf.data_ptr()), buf.numel()
    else:
        buf = np.ascontiguousarray(buf)
        p, n = C.c_void_p(buf.ctypes.data), buf.size
    D = C.c_int()
    shp = (C.c_uint64 * MAX_DIM)()
    _check(L.mgh_infer_shape(p, n, C.byref(D), shp))
    dt = C.c_int()
    _check(L.mgh_infer_data_type(p, n, C.byref(dt)))
    return tuple(int(shp[d]) for d in range(D.value)), dt.value


def decompress(buf, config=None, out=None):
    """mgard_x::decompress. Returns a numpy array (host stream) or a cuda tensor (device stream).
    `out`: optional pre-allocated buffer -- a contiguous cuda tensor (device streams) or a
    C-contiguous numpy array (host streams). Its size and type are checked by the library against the
    header it reads anyway (mgh_decompress_into: ValueError on a mismatch, nothing written)."""
    import torch
    L = _hl()
    cfg = config if config is not None else Config()
    on_dev = isinstance(buf, torch.Tensor) and buf.is_cuda
    if not on_dev:
        buf = np.ascontiguousarray(buf)
    if out is None:
        shape, dt = infer(buf)
        if on_dev:
            out = torch.empty(shape, dtype=torch.float32 if dt == FLOAT else torch.float64, device=buf.device)
        else:
            out = np.empty(shape, dtype=np.float32 if dt == FLOAT else np.float64)
    if on_dev:
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.is_contiguous() and
                out.dtype in (torch.float32, torch.float64) and out.device == buf.device):
            raise ValueError("`out` must be a contiguous float32/float64 cuda tensor on the stream's device")
        p, n, optr = C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(out.data_ptr())
        nbytes, odt = out.numel() * out.element_size(), FLOAT if out.dtype == torch.float32 else DOUBLE
    else:
        if not (isinstance(out, np.ndarray) and out.flags.c_contiguous and out.flags.writeable and
                out.dtype in (np.float32, np.float64)):
            raise ValueError("`out` must be a writeable C-contiguous float32/float64 numpy array")
        p, n, optr = C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(out.ctypes.data)
        nbytes, odt = out.nbytes, FLOAT if out.dtype == np.float32 else DOUBLE
    rc = L.mgh_decompress_into(p, n, optr, nbytes, odt, C.byref(cfg))
    if rc == -1 and b"mgh_decompress_into" in L.mgh_last_error():
        raise ValueError(L.mgh_last_error().decode())
    _check(rc)
    return out


def pin(a):
    """mgard_x::pin_memory on a numpy array (hipHostRegister): transfers of it run asynchronously at
    the link's rate. Undo with unpin() before the array is freed."""
    _check(_hl().mgh_pin_memory(C.c_void_p(a.ctypes.data), a.nbytes))


def is_pinned(a):
    return bool(_hl().mgh_check_memory_pinned(C.c_void_p(a.ctypes.data)))


def unpin(a):
    _check(_hl().mgh_unpin_memory(C.c_void_p(a.ctypes.data)))


def compress_multi(data, tol, s=INF, mode=REL, devices=(0,), coords=None, config=None):
    """mgh_compress_multi: slab id of the slowest dimension runs on devices[id % len(devices)] (one
    host thread per device). `data`: host numpy array -> host stream (numpy uint8), or a cuda
    tensor resident on ONE device -> cuda uint8 tensor on that device (slabs of other devices travel
    device to device)."""
    import torch
    L = _hl()
    cfg = config if config is not None else Config()
    on_device = isinstance(data, torch.Tensor) and data.is_cuda
    if on_device:
        data = data.contiguous()
        np_dt = np.dtype(np.float32) if data.dtype == torch.float32 else np.dtype(np.float64)
        shape, nbytes, dptr = tuple(data.shape), data.numel() * data.element_size(), data.data_ptr()
    else:
        data = np.ascontiguousarray(data)
        np_dt, shape, nbytes, dptr = data.dtype, data.shape, data.nbytes, data.ctypes.data
    dt = {np.dtype(np.float32): FLOAT, np.dtype(np.float64): DOUBLE}[np_dt]
    D = len(shape)
    shp = (C.c_uint64 * D)(*shape)
    cptr, ckeep = None, []
    if coords is not None:
        arr = (C.c_void_p * D)()
        for d in range(D):
            c = np.ascontiguousarray(coords[d], dtype=np_dt)
            ckeep.append(c)
            arr[d] = c.ctypes.data
        cptr = arr
    devs = (C.c_int * len(devices))(*devices)
    cap = nbytes + 1000000
    if on_device:
        out = torch.empty(cap, dtype=torch.uint8, device=data.device)
        optr = C.c_void_p(out.data_ptr())
    else:
        out = np.empty(cap, dtype=np.uint8)
        optr = C.c_void_p(out.ctypes.data)
    size = C.c_size_t(cap)
    _check(L.mgh_compress_multi(len(devices), devs, D, dt, shp, float(tol), float(s), int(mode),
                                C.c_void_p(dptr), C.byref(optr), C.byref(size), cptr,
                                C.byref(cfg), 1))
    return out[:size.value]


def decompress_multi(buf, devices=(0,), config=None):
    """mgh_decompress_multi: host stream in, host numpy array out."""
    L = _hl()
    cfg = config if config is not None else Config()
    shape, dt = infer(buf)
    buf = np.ascontiguousarray(buf)
    out = np.empty(shape, dtype=np.float32 if dt == FLOAT else np.float64)
    optr = C.c_void_p(out.ctypes.data)
    devs = (C.c_int * len(devices))(*devices)
    _check(L.mgh_decompress_multi(len(devices), devs, C.c_void_p(buf.ctypes.data), buf.size,
                                  C.byref(optr), C.byref(cfg), 1))
    return out


def compress_dist(comm, rank, nranks, local, tol, s=INF, mode=REL, root=0, config=None, rccl_path=None):
    """mgh_compress_dist: `comm` = the ncclComm_t (int / c_void_p) of an RCCL communicator of `nranks`
    ranks, `local` = this rank's slab (cuda tensor). Returns the container (cuda uint8 tensor) on the
    root, None elsewhere. rccl_path: the librccl the communicator was created with (when it is not
    the one already in the process / librccl.so.1)."""
    import torch
    L = _hl()
    if rccl_path is not None:
        _check(L.mgh_dist_use_library(str(rccl_path).encode()))
    cfg = config if config is not None else Config()
    local = local.contiguous()
    dt = FLOAT if local.dtype == torch.float32 else DOUBLE
    shp = (C.c_uint64 * local.dim())(*local.shape)
    optr, size = C.c_void_p(), C.c_size_t(0)
    _check(L.mgh_compress_dist(C.c_void_p(int(comm)), rank, nranks, root, local.dim(), dt, shp, float(tol), float(s),
                               int(mode), C.c_void_p(local.data_ptr()), C.byref(optr), C.byref(size), None,
                               C.byref(cfg), 0))
    if rank != root:
        return None
    out = torch.empty(size.value, dtype=torch.uint8, device=local.device)
    _check(L.mgh_memcpy(C.c_void_p(out.data_ptr()), optr, size.value))
    L.mgh_free_device(optr)
    return out


def decompress_dist(comm, rank, nranks, container, local_shape, dtype, root=0, config=None, device=None):
    """mgh_decompress_dist: the root passes the container (cuda uint8 tensor), the others None; every
    rank gets its slab (cuda tensor of local_shape)."""
    import torch
    L = _hl()
    cfg = config if config is not None else Config()
    dev = container.device if container is not None else device
    out = torch.empty(tuple(local_shape), dtype=dtype, device=dev)
    p = C.c_void_p(container.data_ptr()) if container is not None else None
    n = int(container.numel()) if container is not None else 0
    _check(L.mgh_decompress_dist(C.c_void_p(int(comm)), rank, nranks, root, p, n, C.c_void_p(out.data_ptr()),
                                 C.byref(cfg)))
    return out


def release_cache():
    _hl().mgh_release_cache()


# ---- lossless stage on its own ----------------------------------------------------------------
def huffman_codebook(freq):
    """(code, first, entry, keys) for a histogram (host only, no device needed)."""
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    n = f.size
    code, keys = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    first, entry = np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    _check(_hl().mgh_huffman_codebook(f.ctypes.data, n, code.ctypes.data, first.ctypes.data,
                                      entry.ctypes.data, keys.ctypes.data))
    return code, first, entry, keys


class Lossless:
    """mgh_lossless_ctx: Huffman [+ Zstd] on quantized symbols held in device memory."""

    def __init__(self, dev_id=0):
        self._c = C.c_void_p()
        _check(_hl().mgh_lossless_create(C.byref(self._c), dev_id))

    def close(self):
        if self._c:
            _hl().mgh_lossless_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compress(self, q, dict_size=8192, chunk_size=20480, lossless=HUFFMAN, zstd_level=3,
                 outlier_idx=None, outlier_val=None):
        """q: cuda int64 tensor of symbols in [0, dict_size). Returns the payload bytes."""
        import torch
        n_out = 0 if outlier_idx is None else int(outlier_idx.numel())
        pay, size = C.c_void_p(), C.c_uint64()
        _check(_hl().mgh_lossless_compress(
            self._c, C.c_void_p(q.data_ptr()), q.numel(), dict_size, chunk_size, lossless, zstd_level,
            C.c_void_p(outlier_idx.data_ptr()) if n_out else None,
            C.c_void_p(outlier_val.data_ptr()) if n_out else None, n_out, C.byref(pay), C.byref(size),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return C.string_at(pay, size.value)

    def compress_device(self, q, out, dict_size=8192, chunk_size=20480, outlier_idx=None, outlier_val=None):
        """The record written into `out` (cuda uint8 tensor or a view of one: any byte alignment).
        Returns the record as a view of `out`."""
        import torch
        n_out = 0 if outlier_idx is None else int(outlier_idx.numel())
        size = C.c_uint64()
        _check(_hl().mgh_lossless_compress_device(
            self._c, C.c_void_p(q.data_ptr()), q.numel(), dict_size, chunk_size,
            C.c_void_p(outlier_idx.data_ptr()) if n_out else None,
            C.c_void_p(outlier_val.data_ptr()) if n_out else None, n_out, C.c_void_p(out.data_ptr()),
            out.numel(), C.byref(size), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out[:size.value]

    def decompress(self, payload, n, lossless=HUFFMAN):
        """payload: bytes, or a cuda uint8 tensor (decoded where it lies). Returns (q, outlier_idx,
        outlier_val) as cuda tensors."""
        import torch
        q = torch.empty(n, dtype=torch.int64, device="cuda")
        if isinstance(payload, torch.Tensor):
            if not (payload.is_cuda and payload.dtype == torch.uint8 and payload.dim() == 1 and
                    payload.is_contiguous()):
                raise ValueError("payload tensor must be a contiguous 1-D uint8 cuda tensor")
            if payload.device != q.device:
                raise ValueError("payload tensor is on %s, the context decodes on %s" % (payload.device, q.device))
            raw, nbytes = C.c_void_p(payload.data_ptr()), int(payload.numel())
        else:
            raw, nbytes = (C.c_uint8 * len(payload)).from_buffer_copy(payload), len(payload)
        oi, ov, cnt = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(_hl().mgh_lossless_decompress(
            self._c, raw, nbytes, lossless, C.c_void_p(q.data_ptr()), n, C.byref(oi), C.byref(ov),
            C.byref(cnt), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        k = cnt.value
        idx = torch.empty(k, dtype=torch.int64, device="cuda")
        val = torch.empty(k, dtype=torch.int64, device="cuda")
        if k:
            _check(_hl().mgh_memcpy(C.c_void_p(idx.data_ptr()), oi, k * 8))
            _check(_hl().mgh_memcpy(C.c_void_p(val.data_ptr()), ov, k * 8))
        return q, idx, val
