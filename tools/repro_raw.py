import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch, mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
rng = np.random.default_rng(1)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    for shape in [(40, 41, 42), (5, 12, 13, 14)]:
        u = rng.standard_normal(shape).astype(np.float32) * 1e6
        buf = hl.compress(u, 1e-9, np.inf, mg.ABS)
        v = hl.decompress(buf)
        assert np.array_equal(v, u)
    if it % 10 == 0:
        w = smooth_field((65, 70, 129), np.float32)
        b2 = hl.compress(w, 1e-3, np.inf, mg.REL)
        hl.decompress(b2)
    if it % 50 == 0:
        print("iter", it, flush=True)
print("done")
