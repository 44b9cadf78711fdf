"""TEST INFRASTRUCTURE -- ctypes binding of the CPU oracle (oracle/mgx_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / reported CPU baseline.
The product package (``mgard_amd``) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmgx_oracle.so")
_lib = None
# libgomp spin-waits by default; with many tiny parallel regions (unit tests) that burns the
# 8 build-container cores. Must be set before libgomp initialises.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

REL, ABS = 0, 1  # mgard_x::error_bound_type (Utilities/Types.h:32)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("mgx_oracle.c", "mgx_oracle_impl.h", "mgcpu_1d.c")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmgx_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_SFX = {np.dtype(np.float32): ("_f32", C.c_float), np.dtype(np.float64): ("_f64", C.c_double)}


def _declare(L):
    u64p = C.POINTER(C.c_uint64)
    i64p = C.POINTER(C.c_int64)
    for sfx, ct in _SFX.values():
        rp = C.POINTER(ct)
        f = getattr(L, "mgxo_hier_create" + sfx)
        f.restype = C.c_void_p
        f.argtypes = [C.c_int, u64p, C.POINTER(rp), C.c_int, C.c_uint64]
        getattr(L, "mgxo_hier_destroy" + sfx).argtypes = [C.c_void_p]
        getattr(L, "mgxo_hier_destroy" + sfx).restype = None
        getattr(L, "mgxo_l_target" + sfx).argtypes = [C.c_void_p]
        getattr(L, "mgxo_l_target" + sfx).restype = C.c_int
        getattr(L, "mgxo_level_shape" + sfx).argtypes = [C.c_void_p, C.c_int, u64p]
        getattr(L, "mgxo_level_shape" + sfx).restype = None
        for name in ("dist", "ratio", "am", "bm"):
            g = getattr(L, "mgxo_" + name + sfx)
            g.argtypes = [C.c_void_p, C.c_int, C.c_int]
            g.restype = rp
        g = getattr(L, "mgxo_marks" + sfx)
        g.argtypes = [C.c_void_p, C.c_int]
        g.restype = C.POINTER(C.c_int)
        for name in ("mgxo_decompose", "mgxo_recompose", "mgxo_decompose_nd", "mgxo_recompose_nd"):
            g = getattr(L, name + sfx)
            g.argtypes = [C.c_void_p, rp]
            g.restype = C.c_int
        for name in ("mgxo_op_lerp", "mgxo_op_mass_trans"):
            g = getattr(L, name + sfx)
            g.argtypes = [C.c_void_p, C.c_int, C.c_int, rp, rp]
            g.restype = C.c_int
        g = getattr(L, "mgxo_op_thomas" + sfx)
        g.argtypes = [C.c_void_p, C.c_int, C.c_int, rp]
        g.restype = C.c_int
        g = getattr(L, "mgxo_calc_quantizers" + sfx)
        g.argtypes = [C.c_void_p, C.c_int, ct, ct, ct, C.c_int, rp]
        g.restype = None
        g = getattr(L, "mgxo_quantize" + sfx)
        g.argtypes = [C.c_void_p, rp, C.c_int, ct, ct, ct, C.c_uint64, C.c_int, i64p, u64p, i64p,
                      C.c_uint64]
        g.restype = C.c_uint64
        g = getattr(L, "mgxo_dequantize" + sfx)
        g.argtypes = [C.c_void_p, i64p, C.c_int, ct, ct, ct, C.c_uint64, C.c_int, u64p, i64p,
                      C.c_uint64, rp]
        g.restype = None
        g = getattr(L, "mgxo_norm" + sfx)
        g.argtypes = [rp, C.c_uint64, ct, C.c_int]
        g.restype = ct
        g = getattr(L, "mgxo_linearized_position" + sfx)
        g.argtypes = [C.c_void_p, C.c_uint64]
        g.restype = C.c_uint64
        g = getattr(L, "mgxo_level_linearize" + sfx)
        g.argtypes = [C.c_void_p, i64p, i64p, C.c_int]
        g.restype = None
    L.mgxo_num_threads.restype = C.c_int
    L.mgxo_set_num_threads.argtypes = [C.c_int]


def num_threads():
    return lib().mgxo_num_threads()


def set_num_threads(n):
    lib().mgxo_set_num_threads(int(n))


class Hierarchy:
    """Oracle counterpart of mgard_x::Hierarchy<D,T,SERIAL> (Hierarchy.hpp:193-418)."""

    def __init__(self, shape, dtype=np.float32, coords=None, normalize_coordinates=True,
                 max_level=2**62):
        self.dtype = np.dtype(dtype)
        self.sfx, self.ct = _SFX[self.dtype]
        self.shape = tuple(int(s) for s in shape)
        self.D = len(self.shape)
        L = lib()
        shp = (C.c_uint64 * self.D)(*self.shape)
        cptr = None
        if coords is not None:
            self._coords = [np.ascontiguousarray(c, dtype=self.dtype) for c in coords]
            assert len(self._coords) == self.D
            for c, n in zip(self._coords, self.shape):
                assert c.shape == (n,)
            arr = (C.POINTER(self.ct) * self.D)(
                *[c.ctypes.data_as(C.POINTER(self.ct)) for c in self._coords])
            cptr = arr
        self._h = getattr(L, "mgxo_hier_create" + self.sfx)(self.D, shp, cptr,
                                                           int(normalize_coordinates),
                                                           int(max_level))
        if not self._h:
            raise ValueError("invalid shape for mgard_x hierarchy: %r" % (self.shape,))
        self.l_target = getattr(L, "mgxo_l_target" + self.sfx)(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(lib(), "mgxo_hier_destroy" + self.sfx)(self._h)
            self._h = None

    def level_shape(self, l):
        out = (C.c_uint64 * self.D)()
        getattr(lib(), "mgxo_level_shape" + self.sfx)(self._h, l, out)
        return tuple(int(x) for x in out)

    def _arr(self, name, l, d, n):
        p = getattr(lib(), "mgxo_" + name + self.sfx)(self._h, l, d)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def dist(self, l, d):
        return self._arr("dist", l, d, self.level_shape(l)[d])

    def ratio(self, l, d):
        return self._arr("ratio", l, d, self.level_shape(l)[d])

    def am(self, l, d):
        return self._arr("am", l, d, self.level_shape(l)[d] + 1)

    def bm(self, l, d):
        return self._arr("bm", l, d, self.level_shape(l)[d] + 1)

    def marks(self, d):
        p = getattr(lib(), "mgxo_marks" + self.sfx)(self._h, d)
        return np.ctypeslib.as_array(p, shape=(self.shape[d],)).copy()

    def _rp(self, a):
        return a.ctypes.data_as(C.POINTER(self.ct))

    def decompose(self, data, force_nd=False):
        """Returns multilevel coefficients in MGARD-X's in-place reordered layout. force_nd
        runs the generic N-D restatement (the D > 3 path) also for D <= 3."""
        v = np.array(data, dtype=self.dtype, order="C", copy=True).reshape(self.shape)
        fn = "mgxo_decompose_nd" if force_nd else "mgxo_decompose"
        rc = getattr(lib(), fn + self.sfx)(self._h, self._rp(v))
        if rc:
            raise NotImplementedError("oracle decompose: D=%d unsupported" % self.D)
        return v

    def recompose(self, coeffs, force_nd=False):
        v = np.array(coeffs, dtype=self.dtype, order="C", copy=True).reshape(self.shape)
        fn = "mgxo_recompose_nd" if force_nd else "mgxo_recompose"
        rc = getattr(lib(), fn + self.sfx)(self._h, self._rp(v))
        if rc:
            raise NotImplementedError("oracle recompose: D=%d unsupported" % self.D)
        return v

    # ---- the line operators on their own (test hooks; odd level sizes) ----
    def op_lerp(self, l, d, fine):
        """Interpolants at the odd nodes of level l along dimension d (GridProcessingKernel3D.hpp:614-617)."""
        f = np.ascontiguousarray(fine, dtype=self.dtype)
        n = self.level_shape(l)[d]
        assert f.shape == (n,)
        out = np.zeros(n // 2, dtype=self.dtype)
        assert getattr(lib(), "mgxo_op_lerp" + self.sfx)(self._h, l, d, self._rp(f), self._rp(out)) == 0
        return out

    def op_mass_trans(self, l, d, fine):
        """Restriction of the level-l mass matrix applied to `fine`, on the nodes of level l - 1
        (LPKFunctor.h:77-93 through one Lpk line)."""
        f = np.ascontiguousarray(fine, dtype=self.dtype)
        assert f.shape == (self.level_shape(l)[d],)
        out = np.zeros(self.level_shape(l - 1)[d], dtype=self.dtype)
        assert getattr(lib(), "mgxo_op_mass_trans" + self.sfx)(self._h, l, d, self._rp(f), self._rp(out)) == 0
        return out

    def op_thomas(self, l, d, rhs):
        """Solution of the level-l mass system along dimension d (IPKFunctor.h:111-149)."""
        x = np.array(rhs, dtype=self.dtype, order="C", copy=True)
        assert x.shape == (self.level_shape(l)[d],)
        assert getattr(lib(), "mgxo_op_thomas" + self.sfx)(self._h, l, d, self._rp(x)) == 0
        return x

    def quantizers(self, ebtype, tol, s, norm, reciprocal):
        out = np.zeros(self.l_target + 1, dtype=self.dtype)
        getattr(lib(), "mgxo_calc_quantizers" + self.sfx)(self._h, ebtype, tol, s, norm,
                                                         int(reciprocal), self._rp(out))
        return out

    def quantize(self, coeffs, ebtype, tol, s, norm, dict_size=8192, prep_huffman=True,
                 outlier_cap=None):
        v = np.ascontiguousarray(coeffs, dtype=self.dtype).reshape(self.shape)
        n = v.size
        cap = n if outlier_cap is None else int(outlier_cap)
        q = np.empty(self.shape, dtype=np.int64)
        oi = np.empty(max(cap, 1), dtype=np.uint64)
        ov = np.empty(max(cap, 1), dtype=np.int64)
        cnt = getattr(lib(), "mgxo_quantize" + self.sfx)(
            self._h, self._rp(v), ebtype, tol, s, norm, dict_size, int(prep_huffman),
            q.ctypes.data_as(C.POINTER(C.c_int64)), oi.ctypes.data_as(C.POINTER(C.c_uint64)),
            ov.ctypes.data_as(C.POINTER(C.c_int64)), cap)
        k = min(int(cnt), cap)
        return q, oi[:k].copy(), ov[:k].copy(), int(cnt)

    def dequantize(self, q, ebtype, tol, s, norm, dict_size=8192, prep_huffman=True,
                   outlier_idx=None, outlier_val=None):
        qq = np.array(q, dtype=np.int64, order="C", copy=True).reshape(self.shape)
        oi = np.ascontiguousarray(outlier_idx if outlier_idx is not None else [], dtype=np.uint64)
        ov = np.ascontiguousarray(outlier_val if outlier_val is not None else [], dtype=np.int64)
        out = np.empty(self.shape, dtype=self.dtype)
        getattr(lib(), "mgxo_dequantize" + self.sfx)(
            self._h, qq.ctypes.data_as(C.POINTER(C.c_int64)), ebtype, tol, s, norm, dict_size,
            int(prep_huffman), oi.ctypes.data_as(C.POINTER(C.c_uint64)),
            ov.ctypes.data_as(C.POINTER(C.c_int64)), len(oi), self._rp(out))
        return out


def _level_linearize(self, q, inverse=False):
    """config.reorder == 1: the quantized array level by level (LinearQuantization.hpp:46-146,
    588-605); inverse=True undoes it. Returns a flat int64 array (forward) / an array of the
    hierarchy's shape (inverse)."""
    a = np.ascontiguousarray(q, dtype=np.int64).reshape(-1)
    out = np.empty_like(a)
    getattr(lib(), "mgxo_level_linearize" + self.sfx)(
        self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_int64)),
        int(inverse))
    return out.reshape(self.shape) if inverse else out


def _linearized_position(self, lin):
    return int(getattr(lib(), "mgxo_linearized_position" + self.sfx)(self._h, int(lin)))


Hierarchy.level_linearize = _level_linearize
Hierarchy.linearized_position = _linearized_position


def norm(data, s, normalize_coordinates=True):
    v = np.ascontiguousarray(data)
    sfx, ct = _SFX[v.dtype]
    return float(getattr(lib(), "mgxo_norm" + sfx)(v.ctypes.data_as(C.POINTER(ct)), v.size, s,
                                                  int(normalize_coordinates)))


def dyadic_natural_to_reordered(a):
    """Permute an array given in natural node order on a dyadic grid (every dim 2^k+1)
    into MGARD-X's in-place reordered layout (coarse corner first, level by level). A node's
    level is the max over dims of its per-dim level, and ALL its coordinates are laid out by
    that level's split (coarse index p/2 for even p, n_coarse + (p-1)/2 for odd p), which is why
    this is not a tensor product of 1-D permutations. Used to compare with the reference's
    MGARD-CPU goldens, which are in natural order after mgard::unshuffle."""
    a = np.asarray(a)
    idx = _dyadic_maps(a.shape)
    out = np.empty_like(a)
    out[idx] = a
    return out


def dyadic_reordered_to_natural(a):
    a = np.asarray(a)
    return a[_dyadic_maps(a.shape)]


def _dyadic_maps(shape):
    """Every dim 2^k + 1 (k may differ between the dims; the hierarchy has L = min k levels)."""
    ks = [(n - 1).bit_length() - 1 for n in shape]
    assert all(n == (1 << k) + 1 and n >= 2 for n, k in zip(shape, ks)), "dyadic sizes only"
    L = min(ks)
    nat = np.indices(shape)  # natural coordinates of every node
    # per-dim level of coordinate p: smallest l with p % 2^(L-l) == 0
    lev = []
    for n in shape:
        lev1 = np.zeros(n, dtype=np.int64)
        for p in range(n):
            l = 0
            while p % (1 << (L - l)) != 0:
                l += 1
            lev1[p] = l
        lev.append(lev1)
    level = np.max(np.stack([lev[d][nat[d]] for d in range(len(shape))]), axis=0)
    out = []
    for d, n in enumerate(shape):
        pl = nat[d] >> (L - level)          # coordinate on the node's own level grid
        ncoarse = np.where(level > 0, ((n - 1) >> (L - np.maximum(level, 1) + 1)) + 1, 0)
        out.append(np.where(level == 0, pl, np.where(pl % 2 == 0, pl // 2, ncoarse + (pl - 1) // 2)))
    return tuple(out)


# ---- MGARD-CPU (mgard::compress, BASELINE.json configs[0]) restated for one dimension -------
class MgardCpu1D:
    """oracle/mgcpu_1d.c: hierarchy, decompose / recompose (natural node order) and the s = inf
    quantizer of the serial CPU code path, float64."""

    def __init__(self, n, coords=None):
        self.n = int(n)
        self.x = (np.arange(self.n, dtype=np.float64) / (self.n - 1) if coords is None
                  else np.ascontiguousarray(coords, dtype=np.float64))
        L = lib()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.mgcpu1d_levels.argtypes = [C.c_uint64]
        L.mgcpu1d_level_size.argtypes = [C.c_uint64, C.c_int]
        L.mgcpu1d_level_size.restype = C.c_uint64
        L.mgcpu1d_decompose.argtypes = [C.c_uint64, dp, dp]
        L.mgcpu1d_recompose.argtypes = [C.c_uint64, dp, dp]
        L.mgcpu1d_quantum.argtypes = [C.c_uint64, C.c_double]
        L.mgcpu1d_quantum.restype = C.c_double
        L.mgcpu1d_quantize.argtypes = [C.c_uint64, dp, C.c_double, ip]
        L.mgcpu1d_dequantize.argtypes = [C.c_uint64, ip, C.c_double, dp]
        self.L = L.mgcpu1d_levels(self.n)

    def _dp(self, a):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    def level_size(self, l):
        return int(lib().mgcpu1d_level_size(self.n, l))

    def decompose(self, u):
        v = np.array(u, dtype=np.float64, order="C", copy=True)
        lib().mgcpu1d_decompose(self.n, self._dp(self.x), self._dp(v))
        return v

    def recompose(self, c):
        v = np.array(c, dtype=np.float64, order="C", copy=True)
        lib().mgcpu1d_recompose(self.n, self._dp(self.x), self._dp(v))
        return v

    def quantum(self, tol):
        return float(lib().mgcpu1d_quantum(self.n, float(tol)))

    def quantize(self, v, quantum):
        v = np.ascontiguousarray(v, dtype=np.float64)
        q = np.empty(self.n, dtype=np.int64)
        lib().mgcpu1d_quantize(self.n, self._dp(v), float(quantum), q.ctypes.data_as(C.POINTER(C.c_int64)))
        return q

    def dequantize(self, q, quantum):
        q = np.ascontiguousarray(q, dtype=np.int64)
        v = np.empty(self.n, dtype=np.float64)
        lib().mgcpu1d_dequantize(self.n, q.ctypes.data_as(C.POINTER(C.c_int64)), float(quantum), self._dp(v))
        return v
