import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from tests.util import smooth_field
for shape in [(400, 30, 50), (40, 330, 70), (12, 320, 9, 33)]:
    u = smooth_field(shape, np.float32, noise=3e-3)
    h = mg.Hierarchy(shape, np.float32)
    c = h.decompose(torch.from_numpy(u).cuda()); b = h.recompose(c)
    torch.cuda.synchronize(); h.close()
