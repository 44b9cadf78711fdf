"""Per-kernel times of the step with int64 output vs 16-bit symbol output (same box, same input)."""
import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
n1 = int(os.environ.get("NSIDE", "512"))
shape = (n1, n1, n1)
u = smooth_field(shape, np.float32)
d = torch.from_numpy(u).cuda()
h = mgard_amd.Hierarchy(shape, np.float32)
N = n1 ** 3
cap = N // 16
q = torch.empty(shape, dtype=torch.int64, device='cuda')
sym = torch.empty(shape, dtype=torch.uint16, device='cuda')
cnt = torch.zeros(1, dtype=torch.int64, device='cuda')
oidx = torch.empty(cap, dtype=torch.int64, device='cuda'); oval = torch.empty(cap, dtype=torch.int64, device='cuda')
lib = mgard_amd.load_library()
def run64():
    h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=(q, cnt, oidx, oval), want_norm=False)
def run16():
    lib.mgh_decompose_quantize_sym16(h._h, C.c_void_p(d.data_ptr()), mgard_amd.REL, 1e-3, float("inf"), 0.0, None, 8192,
                                     C.c_void_p(sym.data_ptr()), C.c_void_p(cnt.data_ptr()), C.c_void_p(oidx.data_ptr()),
                                     C.c_void_p(oval.data_ptr()), cap, None)
for name, f in (("int64", run64), ("sym16", run16)) * 2:
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 20 * 1e3
    h.profile(True)
    for _ in range(10): f()
    torch.cuda.synchronize()
    pr = h.profile_read()
    h.profile(False)
    print(name, "%.4f ms" % ms, {k: round(v[0] / 10 * 1000) for k, v in sorted(pr.items())})
