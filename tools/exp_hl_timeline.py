import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
u = smooth_field((512, 512, 512), np.float32); ud = torch.from_numpy(u).cuda()
for i in range(3):
    print("---- compress", i, file=sys.stderr); buf = hl.compress(ud, 1e-3, np.inf, mg.REL)
for i in range(2):
    print("---- decompress", i, file=sys.stderr); v = hl.decompress(buf)
