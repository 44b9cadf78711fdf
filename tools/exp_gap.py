"""Do the HIP events around the dominant kernel cost the step anything? 6 steps without them, 6 with
(rocprofv3 --kernel-trace -- python3 tools/exp_gap.py; tools/timeline.py db --steps-back 1 / 7)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
shape = (512, 512, 512)
u = smooth_field(shape, np.float32); d = torch.from_numpy(u).cuda()
h = mgard_amd.Hierarchy(shape, np.float32)
cap = u.size // 16
bufs = (torch.empty(shape, dtype=torch.int64, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda'),
        torch.empty(cap, dtype=torch.int64, device='cuda'), torch.empty(cap, dtype=torch.int64, device='cuda'))
import time
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
run(3)
print("no events      %.4f ms/step" % run(20))
h.profile(True, only="level_fused_q")
print("events on pass %.4f ms/step" % run(20))
h.profile_read(reset=True)
h.profile(False)
print("no events      %.4f ms/step" % run(20))
