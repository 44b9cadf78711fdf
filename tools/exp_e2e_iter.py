import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
u = smooth_field((512, 512, 512), np.float32); ud = torch.from_numpy(u).cuda()
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf = hl.compress(ud, 1e-3, np.inf, mg.REL)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    v = hl.decompress(buf)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("iter %d compress %.2f ms decompress %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
