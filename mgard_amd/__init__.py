"""mgard_amd -- MI355X-native MGARD-X hot path (multilevel decomposition + level-wise linear
quantizer) behind the C ABI of include/mgard_hip.h.

This module is the thin Python host side used by the tests and bench.py: torch supplies device
memory and streams, every computation goes through libmgard_hip.so (hand-written HIP kernels).
There is no CPU fallback: without the built library or without a GPU the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import _build

REL, ABS = 0, 1          # mgard_x::error_bound_type
LD_IN, LD_OUT = 0, 1     # mgh_set_ld
FLOAT, DOUBLE = 0, 1     # mgard_x::data_type
INF = float("inf")

_lib = None

SYMBOLS = [
    "mgh_last_error", "mgh_device_count", "mgh_hierarchy_create", "mgh_hierarchy_destroy",
    "mgh_l_target", "mgh_level_shape", "mgh_total_num_elems", "mgh_device_bytes", "mgh_norm_device_ptr",
    "mgh_hierarchy_table", "mgh_norm", "mgh_decompose", "mgh_recompose", "mgh_quantize",
    "mgh_dequantize", "mgh_decompose_quantize", "mgh_dequantize_recompose",
    "mgh_norm_device", "mgh_decompose_quantize_dn", "mgh_decompose_quantize_sym16",
    "mgh_dequantize_recompose_sym16", "mgh_sym16_supported",
    "mgh_profile_enable", "mgh_profile_filter", "mgh_profile_read", "mgh_stream_calibrate",
    "mgh_level_linearize", "mgh_outlier_restore", "mgh_norm_stream_begin", "mgh_norm_stream_add",
    "mgh_set_ld",
]


class MgardHipError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def load_library():
    """Loads libmgard_hip.so (importing torch first so that both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_build.LIB):
        raise MgardHipError(
            "libmgard_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    import torch  # noqa: F401  (loads libamdhip64.so.7 that our library binds to)
    L = C.CDLL(_build.LIB)
    vp, u64, i64p, u64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    L.mgh_last_error.restype = C.c_char_p
    L.mgh_device_count.restype = C.c_int
    L.mgh_hierarchy_create.argtypes = [C.POINTER(vp), C.c_int, u64p, C.c_int, C.POINTER(vp),
                                       C.c_int, u64, C.c_int]
    L.mgh_hierarchy_destroy.argtypes = [vp]
    L.mgh_hierarchy_destroy.restype = None
    L.mgh_l_target.argtypes = [vp]
    L.mgh_level_shape.argtypes = [vp, C.c_int, u64p]
    L.mgh_total_num_elems.argtypes = [vp]
    L.mgh_total_num_elems.restype = u64
    L.mgh_device_bytes.argtypes = [vp]
    L.mgh_device_bytes.restype = C.c_size_t
    L.mgh_hierarchy_table.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, u64]
    L.mgh_hierarchy_table.restype = C.c_int64
    L.mgh_norm.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_double), vp]
    L.mgh_decompose.argtypes = [vp, vp, vp, vp]
    L.mgh_recompose.argtypes = [vp, vp, vp, vp]
    L.mgh_quantize.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double, u64, C.c_int,
                               vp, vp, vp, vp, u64, vp]
    L.mgh_dequantize.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double, u64,
                                 C.c_int, vp, vp, u64, vp, vp]
    L.mgh_decompose_quantize.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double,
                                         C.POINTER(C.c_double), u64, C.c_int, vp, vp, vp, vp, u64,
                                         vp, vp]
    L.mgh_norm_device.argtypes = [vp, vp, C.c_double, vp, vp]
    L.mgh_norm_stream_begin.argtypes = [vp, vp]
    L.mgh_set_ld.argtypes = [vp, C.c_int, u64p]
    L.mgh_norm_stream_add.argtypes = [vp, vp, u64, C.c_double, C.c_int, vp]
    L.mgh_dequantize_recompose_sym16.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double, u64,
                                                 vp, vp, u64, vp, vp]
    L.mgh_sym16_supported.argtypes = [vp]
    L.mgh_decompose_quantize_sym16.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double,
                                               C.POINTER(C.c_double), u64, vp, vp, vp, vp, u64, vp]
    L.mgh_decompose_quantize_dn.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, vp, u64, u64,
                                            C.c_int, vp, vp, vp, vp, u64, vp]
    L.mgh_dequantize_recompose.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double,
                                           u64, C.c_int, vp, vp, u64, vp, vp]
    L.mgh_profile_enable.argtypes = [vp, C.c_int]
    L.mgh_profile_filter.argtypes = [vp, C.c_char_p]
    L.mgh_profile_read.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), u64p,
                                   C.c_int, C.c_int]
    L.mgh_outlier_restore.argtypes = [vp, u64, vp, vp, u64, vp]
    L.mgh_level_linearize.argtypes = [vp, vp, vp, C.c_int, vp, vp, u64, u64, vp]
    L.mgh_stream_calibrate.argtypes = [C.c_int, vp, vp, vp, u64, C.c_int, C.POINTER(C.c_double), vp]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise MgardHipError("mgard_hip error %d: %s" % (rc, load_library().mgh_last_error().decode()))
    return rc


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_NP = {FLOAT: np.float32, DOUBLE: np.float64}


class Hierarchy:
    """Host mirror of mgard_x::Hierarchy<D,T,HIP> + the stages of mgard_x::Compressor<D,T,HIP>
    (reference include/mgard-x/CompressionLowLevel/Compressor.h:39-78): norm, decompose,
    quantize, dequantize, recompose on torch device tensors."""

    def __init__(self, shape, dtype="float32", coords=None, normalize_coordinates=True,
                 max_level=None, device=0):
        import torch
        L = load_library()
        if not torch.cuda.is_available():
            raise MgardHipError("no HIP device visible: mgard_amd has no CPU fallback")
        self.shape = tuple(int(s) for s in shape)
        self.D = len(self.shape)
        self.np_dtype = np.dtype(dtype)
        self.dtype = {np.dtype(np.float32): FLOAT, np.dtype(np.float64): DOUBLE}[self.np_dtype]
        self.torch_dtype = torch.float32 if self.dtype == FLOAT else torch.float64
        self.device = int(device)
        shp = (C.c_uint64 * self.D)(*self.shape)
        cptr = None
        if coords is not None:
            self._coords = [np.ascontiguousarray(c, dtype=self.np_dtype) for c in coords]
            cptr = (C.c_void_p * self.D)(*[c.ctypes.data for c in self._coords])
        h = C.c_void_p()
        _check(L.mgh_hierarchy_create(C.byref(h), self.D, shp, self.dtype, cptr,
                                      int(normalize_coordinates),
                                      2**64 - 1 if max_level is None else int(max_level),
                                      self.device))
        self._h = h
        self._ld = {LD_IN: None, LD_OUT: None}
        self.l_target = L.mgh_l_target(h)
        self.total = int(L.mgh_total_num_elems(h))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.mgh_hierarchy_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close

    # ---- introspection ----
    def level_shape(self, l):
        out = (C.c_uint64 * self.D)()
        _check(load_library().mgh_level_shape(self._h, l, out))
        return tuple(int(x) for x in out)

    def device_bytes(self):
        return int(load_library().mgh_device_bytes(self._h))

    def table(self, kind, level, dim):
        kinds = {"dist": 0, "ratio": 1, "am": 2, "bm": 3, "marks": 4}
        n = max(self.shape) + 1
        buf = np.zeros(n, dtype=np.int32 if kind == "marks" else self.np_dtype)
        k = _check(load_library().mgh_hierarchy_table(self._h, kinds[kind], level, dim,
                                                      buf.ctypes.data, n))
        return buf[:k].copy()

    # ---- stages ----
    def _chk(self, t, dtype=None):
        import torch
        ok = {self.total}
        if dtype is None:  # (a T array may be a pitched allocation: mgh_set_ld)
            for ld in self._ld.values():
                if ld is not None:
                    n = self.shape[0]
                    for x in ld[1:]:
                        n *= x
                    ok.add(n)
        assert t.is_cuda and t.is_contiguous() and t.numel() in ok, "bad tensor"
        assert t.dtype == (dtype or self.torch_dtype), "bad dtype"
        assert t.device.index == self.device
        return C.c_void_p(t.data_ptr())

    def norm(self, data, s=INF):
        out = C.c_double()
        _check(load_library().mgh_norm(self._h, self._chk(data), s, C.byref(out), _stream()))
        return out.value

    def _new_out(self, device):
        """A T array for the calls to write: dense, or the pitched allocation LD_OUT describes."""
        import torch
        ld = self._ld[LD_OUT]
        shape = self.shape if ld is None else (self.shape[0],) + tuple(ld[1:])
        return torch.empty(shape, dtype=self.torch_dtype, device=device)

    def decompose(self, data, out=None):
        out = self._new_out(data.device) if out is None else out
        _check(load_library().mgh_decompose(self._h, self._chk(data), self._chk(out), _stream()))
        return out

    def recompose(self, coeff, out=None):
        out = self._new_out(coeff.device) if out is None else out
        _check(load_library().mgh_recompose(self._h, self._chk(coeff), self._chk(out), _stream()))
        return out

    def _outlier_bufs(self, cap):
        import torch
        dev = torch.device("cuda", self.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        idx = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
        val = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
        return cnt, idx, val

    def quantize(self, coeff, ebtype, tol, s, norm, dict_size=8192, prep_huffman=True,
                 outlier_cap=None, out=None):
        """Returns (q int64 tensor, outlier_idx, outlier_val, outlier_count)."""
        import torch
        cap = self.total if outlier_cap is None else int(outlier_cap)
        q = torch.empty(self.shape, dtype=torch.int64, device=coeff.device) if out is None else out
        cnt, idx, val = self._outlier_bufs(cap)
        _check(load_library().mgh_quantize(
            self._h, self._chk(coeff), ebtype, tol, s, norm, dict_size, int(prep_huffman),
            self._chk(q, torch.int64), C.c_void_p(cnt.data_ptr()), C.c_void_p(idx.data_ptr()),
            C.c_void_p(val.data_ptr()), cap, _stream()))
        n = int(cnt.item())
        k = min(n, cap)
        return q, idx[:k], val[:k], n

    def dequantize(self, q, ebtype, tol, s, norm, dict_size=8192, prep_huffman=True,
                   outlier_idx=None, outlier_val=None, out=None):
        import torch
        out = self._new_out(q.device) if out is None else out
        n = 0 if outlier_idx is None else int(outlier_idx.numel())
        _check(load_library().mgh_dequantize(
            self._h, self._chk(q, torch.int64), ebtype, tol, s, norm, dict_size, int(prep_huffman),
            C.c_void_p(outlier_idx.data_ptr() if n else 0),
            C.c_void_p(outlier_val.data_ptr() if n else 0), n, self._chk(out), _stream()))
        return out

    def norm_device(self, data, s=INF, out=None):
        """Asynchronous norm: returns a 1-element device tensor of the hierarchy's dtype."""
        import torch
        out = torch.empty(1, dtype=self.torch_dtype, device=data.device) if out is None else out
        _check(load_library().mgh_norm_device(self._h, self._chk(data), s,
                                              C.c_void_p(out.data_ptr()), _stream()))
        return out

    def decompose_quantize_dn(self, data, ebtype, tol, s, d_norm, num_subdomains, bufs,
                              dict_size=8192, prep_huffman=True):
        """Fused hot path with a device-resident GLOBAL norm (decomposed domain); fully async."""
        import torch
        q, cnt, idx, val = bufs
        _check(load_library().mgh_decompose_quantize_dn(
            self._h, self._chk(data), ebtype, tol, s, C.c_void_p(d_norm.data_ptr()),
            int(num_subdomains), dict_size, int(prep_huffman), self._chk(q, torch.int64),
            C.c_void_p(cnt.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(val.data_ptr()),
            int(idx.numel()), _stream()))
        return q, idx, val, cnt

    def decompose_quantize(self, data, ebtype, tol, s, norm=0.0, dict_size=8192,
                           prep_huffman=True, outlier_cap=None, bufs=None, coeff_out=None,
                           want_norm=True):
        """The fused hot path (Compressor::Compress up to the lossless stage). Returns
        (q, outlier_idx, outlier_val, outlier_count, norm). `bufs` = (q, cnt, idx, val) lets a
        caller reuse output buffers; then no host sync happens and outlier_count is the device
        tensor."""
        import torch
        cap = self.total if outlier_cap is None else int(outlier_cap)
        if bufs is None:
            q = torch.empty(self.shape, dtype=torch.int64, device=data.device)
            cnt, idx, val = self._outlier_bufs(cap)
        else:
            q, cnt, idx, val = bufs
            cap = int(idx.numel())
        nout = C.c_double()
        _check(load_library().mgh_decompose_quantize(
            self._h, self._chk(data), ebtype, tol, s, norm,
            C.byref(nout) if want_norm else None, dict_size,
            int(prep_huffman), self._chk(q, torch.int64), C.c_void_p(cnt.data_ptr()),
            C.c_void_p(idx.data_ptr()), C.c_void_p(val.data_ptr()), cap,
            C.c_void_p(coeff_out.data_ptr()) if coeff_out is not None else None, _stream()))
        if bufs is not None:
            return q, idx, val, cnt, nout.value
        n = int(cnt.item())
        k = min(n, cap)
        return q, idx[:k], val[:k], n, nout.value

    def sym16_supported(self):
        return bool(load_library().mgh_sym16_supported(self._h))

    def set_ld(self, which, ld):
        """mgh_set_ld: leading dimensions (len D, or None = dense) of the T arrays the calls read
        (which = LD_IN) or write (LD_OUT). Arrays are then passed as tensors whose STORAGE is the
        pitched allocation (any shape with enough elements)."""
        arr = None if ld is None else (C.c_uint64 * len(ld))(*[int(x) for x in ld])
        _check(load_library().mgh_set_ld(self._h, int(which), arr))
        self._ld[which] = None if ld is None else tuple(int(x) for x in ld)

    def norm_stream(self, data, s, parts):
        """mgh_norm_stream_begin + one mgh_norm_stream_add per part of the (flattened) array: the next
        fused decompose_quantize* call with a REL bound and norm = 0 takes the accumulated norm.
        `parts`: element counts that add up to the array."""
        flat = data.reshape(-1)
        assert sum(parts) == flat.numel()
        L = load_library()
        _check(L.mgh_norm_stream_begin(self._h, _stream()))
        off = 0
        for k, cnt in enumerate(parts):
            _check(L.mgh_norm_stream_add(self._h, C.c_void_p(flat.data_ptr() + off * flat.element_size()),
                                         int(cnt), s, int(k + 1 < len(parts)), _stream()))
            off += cnt

    def decompose_quantize_sym16(self, data, ebtype, tol, s, norm=0.0, dict_size=8192, outlier_cap=None):
        """mgh_decompose_quantize_sym16: (symbols uint16, outlier_idx, outlier_val, count, norm)."""
        import torch
        cap = self.total if outlier_cap is None else int(outlier_cap)
        sym = torch.empty(self.shape, dtype=torch.uint16, device=data.device)
        cnt, idx, val = self._outlier_bufs(cap)
        nout = C.c_double()
        _check(load_library().mgh_decompose_quantize_sym16(
            self._h, self._chk(data), ebtype, tol, s, norm, C.byref(nout), dict_size,
            C.c_void_p(sym.data_ptr()), C.c_void_p(cnt.data_ptr()), C.c_void_p(idx.data_ptr()),
            C.c_void_p(val.data_ptr()), cap, _stream()))
        n = int(cnt.item())
        k = min(n, cap)
        return sym, idx[:k], val[:k], n, nout.value

    def dequantize_recompose_sym16(self, sym, ebtype, tol, s, norm, dict_size=8192, outlier_idx=None,
                                   outlier_val=None, out=None):
        import torch
        out = self._new_out(sym.device) if out is None else out
        n = 0 if outlier_idx is None else int(outlier_idx.numel())
        _check(load_library().mgh_dequantize_recompose_sym16(
            self._h, C.c_void_p(sym.data_ptr()), ebtype, tol, s, norm, dict_size,
            C.c_void_p(outlier_idx.data_ptr() if n else 0), C.c_void_p(outlier_val.data_ptr() if n else 0),
            n, self._chk(out), _stream()))
        return out

    def dequantize_recompose(self, q, ebtype, tol, s, norm, dict_size=8192, prep_huffman=True,
                             outlier_idx=None, outlier_val=None, out=None):
        import torch
        out = self._new_out(q.device) if out is None else out
        n = 0 if outlier_idx is None else int(outlier_idx.numel())
        _check(load_library().mgh_dequantize_recompose(
            self._h, self._chk(q, torch.int64), ebtype, tol, s, norm, dict_size, int(prep_huffman),
            C.c_void_p(outlier_idx.data_ptr() if n else 0),
            C.c_void_p(outlier_val.data_ptr() if n else 0), n, self._chk(out), _stream()))
        return out

    def level_linearize(self, q, inverse=False, outlier_idx=None):
        """mgh_level_linearize: the quantized array level by level (config.reorder == 1) or back;
        forward, `outlier_idx` (int64/uint64 device tensor) is rewritten in place."""
        import torch
        out = torch.empty_like(q)
        n = 0 if outlier_idx is None else int(outlier_idx.numel())
        _check(load_library().mgh_level_linearize(
            self._h, self._chk(q, torch.int64), C.c_void_p(out.data_ptr()), int(inverse),
            C.c_void_p(outlier_idx.data_ptr() if n else 0), None, n, 0, _stream()))
        return out

    # ---- per-kernel timing (HIP events on the launch stream) ----
    def profile(self, enable=True, only=None):
        _check(load_library().mgh_profile_filter(self._h, only.encode() if only else None))
        _check(load_library().mgh_profile_enable(self._h, int(enable)))

    def profile_read(self, reset=True):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_uint64 * cap)()
        n = _check(load_library().mgh_profile_read(self._h, names, ms, cnt, cap, int(reset)))
        return {names[i].decode(): (ms[i], int(cnt[i])) for i in range(min(n, cap))}
