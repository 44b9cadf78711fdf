import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
for shape in [(24,129,140),(65,70,129),(24,65,70),(48,129,140),(12,129,140)]:
    u = smooth_field(shape, np.float32)
    b = hl.compress(u, 1e-3, np.inf, mg.REL)
    h = mg.Hierarchy(shape, np.float32)
    q, oi, ov, cnt, n1 = h.decompose_quantize(torch.from_numpy(u).cuda(), mg.REL, 1e-3, np.inf, outlier_cap=u.size)
    qq = q.cpu().numpy() - 4096
    print(shape, "L", h.l_target, "ratio", u.nbytes / b.size, "outliers", cnt, "q range", qq.min(), qq.max(), "distinct", len(np.unique(qq)))
    h.close()
