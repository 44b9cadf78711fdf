"""mgh_compress + mgh_decompress of ONE shape, three times (for a rocprofv3 --kernel-trace --stats run:
which kernel is it that takes the time on an unusual shape). Dev tool.
  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/exp_one_shape.py 9,4194304 [float64]
FIELD=inside in the environment: a field whose values stay inside the dictionary in 4-D / 5-D too."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
shape = tuple(int(x) for x in sys.argv[1].split(","))
dt = np.dtype(sys.argv[2] if len(sys.argv) > 2 else "float32").type
if os.environ.get("FIELD") == "inside":
    # a field that stays inside the dictionary whatever the dimensionality: one period across the
    # extents of 32 and more points, 1 % of that across the short ones, the usual noise
    ax = np.meshgrid(*[np.arange(n, dtype=np.float64) / max(n - 1, 1) for n in shape], indexing="ij", sparse=True)
    g = sum((1.0 if shape[k] >= 32 else 0.01) * np.sin(2 * np.pi * a + 0.3 * k) for k, a in enumerate(ax))
    g = g + 1e-3 * np.random.default_rng(1).uniform(-1, 1, size=shape)
    u = torch.from_numpy(np.ascontiguousarray(g.astype(dt))).cuda()
else:
    u = torch.from_numpy(smooth_field(shape, dt)).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = hl.compress(u, 1e-3, np.inf, mg.REL)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    v = hl.decompress(s)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: compress %.2f ms, decompress %.2f ms, ratio %.2f" % (shape, (t1 - t0) * 1e3, (t2 - t1) * 1e3, u.numel() * u.element_size() / s.numel()))
