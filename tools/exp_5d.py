"""Step time of 5-D and comparable 4-D / 3-D shapes (same element count): where the N-D path stands."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
for shape in ((8, 8, 64, 64, 64), (64, 64, 64, 64), (256, 256, 256)):
    u = smooth_field(shape, np.float32)
    d = torch.from_numpy(u).cuda()
    h = mgard_amd.Hierarchy(shape, np.float32)
    N = u.size; cap = N
    q = torch.empty(shape, dtype=torch.int64, device='cuda'); cnt = torch.zeros(1, dtype=torch.int64, device='cuda')
    oi = torch.empty(cap, dtype=torch.int64, device='cuda'); ov = torch.empty(cap, dtype=torch.int64, device='cuda')
    f = lambda: h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=(q, cnt, oi, ov), want_norm=False)
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
    print(shape, "%.3f ms  %.1f GB/s" % (ms, u.nbytes / ms / 1e6))
    h.close()
