"""Phase times of the fused level kernel (developer build -DMGH_PHASE_TIMING, see kernels_fused2.hpp)."""
import ctypes as C, os, sys, numpy as np, torch
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
out = (C.c_ulonglong * 16)()
for _ in range(3):
    h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
torch.cuda.synchronize()
L.mgh_debug_phase_read(out, 1)
N = 5
for _ in range(N):
    h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
torch.cuda.synchronize()
L.mgh_debug_phase_read(out, 1)
names = ["A halo cells", "barrier 1", "stash+fetch", "B f-sweep", "barrier 2", "C/D sweeps", "A own cells", "A emit"]
for w in range(2):
    v = [out[w * 8 + k] for k in range(8)]
    tot = sum(v)
    print("wave", 0 if w == 0 else 3, {names[k]: "%.1f%%" % (100.0 * v[k] / tot) for k in range(8)}, "cycles/step total", tot // N)
