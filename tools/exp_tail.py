"""Phase times of the tail kernel (developer build -DMGH_PHASE_TIMING, kernels_tail.hpp):
MGARD_HIP_LIB=build_ab/libmgard_tt.so python tools/exp_tail.py [n]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
u = smooth_field(shape, np.float32); d = torch.from_numpy(u).cuda()
h = mgard_amd.Hierarchy(shape, np.float32)
cap = u.size // 8
bufs = (torch.empty(shape, dtype=torch.int64, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda'),
        torch.empty(cap, dtype=torch.int64, device='cuda'), torch.empty(cap, dtype=torch.int64, device='cuda'))
L = mgard_amd.load_library()
out = (C.c_ulonglong * 64)()
for _ in range(3):
    h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
torch.cuda.synchronize()
L.mgh_debug_tail_read(out, 1)
N = 10
for _ in range(N):
    h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
torch.cuda.synchronize()
L.mgh_debug_tail_read(out, 1)
us = [out[k] / N / 100.0 for k in range(64)]   # 100 MHz ticks -> microseconds
print("tables + first level in %.1f us, solves of the level above %.1f us" % (us[0], us[1]))
names = ["coef", "f-sweep", "c-sweep", "r-sweep", "f-solve", "c-solve", "r-solve", "add"]
for li in range(6):
    v = us[2 + 8 * li: 10 + 8 * li]
    if sum(v) == 0:
        break
    print("tail level %d: %.1f us  " % (li, sum(v)) + "  ".join("%s %.1f" % (a, b) for a, b in zip(names, v)))
print("head %.1f us; total %.1f us" % (us[63], sum(us)))
