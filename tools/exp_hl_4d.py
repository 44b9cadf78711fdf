import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from mgard_amd import highlevel as hl
from tests.util import smooth_field
shape = (8, 512, 512, 512)
base = smooth_field(shape[1:], np.float32)
u = np.stack([base * np.float32(1.0 + 0.002 * t) + np.float32(1e-4 * t) for t in range(shape[0])])
d = torch.from_numpy(u).cuda()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf = hl.compress(d, 1e-3, float('inf'), mgard_amd.REL)
    torch.cuda.synchronize(); print("compress ms", (time.perf_counter() - t0) * 1e3, "ratio", u.nbytes / buf.numel(), file=sys.stderr)
