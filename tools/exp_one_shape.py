"""mgh_compress + mgh_decompress of ONE shape, three times (for a rocprofv3 --kernel-trace --stats run:
which kernel is it that takes the time on an unusual shape). Dev tool.
  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/exp_one_shape.py 9,4194304 [float64]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
shape = tuple(int(x) for x in sys.argv[1].split(","))
dt = np.dtype(sys.argv[2] if len(sys.argv) > 2 else "float32").type
u = torch.from_numpy(smooth_field(shape, dt)).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = hl.compress(u, 1e-3, np.inf, mg.REL)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    v = hl.decompress(s)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: compress %.2f ms, decompress %.2f ms, ratio %.2f" % (shape, (t1 - t0) * 1e3, (t2 - t1) * 1e3, u.numel() * u.element_size() / s.numel()))
