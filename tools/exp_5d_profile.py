"""Per-kernel times of the 5-D step (8 x 8 x 64^3 f32). Dev tool. A second line: the same step on a
field that stays in the dictionary (one period across each 64-point extent, 1 % of that across the
8-point ones -- at 1e-3 the 5-D quantizer's bins are 1/976 of the tolerance, eight points of a full
sine leave 8192 of them -- plus the same noise): the kernels' rate without 225 MB of outlier lists."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd
from tests.util import smooth_field
shape = tuple(int(x) for x in sys.argv[1].split("x")) if len(sys.argv) > 1 else (8, 8, 64, 64, 64)
u = smooth_field(shape, np.float32)
d = torch.from_numpy(u).cuda()
h = mgard_amd.Hierarchy(shape, np.float32)
N = u.size
q = torch.empty(shape, dtype=torch.int64, device='cuda'); cnt = torch.zeros(1, dtype=torch.int64, device='cuda')
oi = torch.empty(N, dtype=torch.int64, device='cuda'); ov = torch.empty(N, dtype=torch.int64, device='cuda')
f = lambda: h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=(q, cnt, oi, ov), want_norm=False)
for _ in range(2): f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
print(shape, "%.3f ms  %.1f GB/s" % (ms, u.nbytes / ms / 1e6), "L =", h.l_target, "outliers", int(cnt.item()))
h.profile(True)
for _ in range(3): f()
torch.cuda.synchronize()
for k, v in sorted(h.profile_read(reset=True).items(), key=lambda kv: -kv[1][0]):
    print("  %-22s %8.1f us/step  %4d launches/step" % (k, v[0] / 3 * 1e3, v[1] // 3))

# the same step on a field resolved in all five dimensions
ax = np.meshgrid(*[np.arange(n, dtype=np.float64) / max(n - 1, 1) for n in shape], indexing="ij", sparse=True)
g = sum((1.0 if shape[k] >= 32 else 0.01) * np.sin(2 * np.pi * a + 0.3 * k) for k, a in enumerate(ax)) + 1e-3 * np.random.default_rng(1).uniform(-1, 1, size=shape)
d = torch.from_numpy(np.ascontiguousarray(g.astype(np.float32))).cuda()
h.profile(False)
for _ in range(2): f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
print("field inside the dictionary:", shape, "%.3f ms  %.1f GB/s" % (ms, u.nbytes / ms / 1e6), "outliers", int(cnt.item()))

# the way back on that field: dequantize + recompose (int64 values in, as the reference's interface)
n = int(cnt.item())
qq = q.clone(); out = torch.empty_like(d)
nrm = h.norm(d, float("inf"))
fb = lambda: h.dequantize_recompose(qq, mgard_amd.REL, 1e-3, float("inf"), nrm, outlier_idx=oi[:n], outlier_val=ov[:n], out=out)
for _ in range(2): fb()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): fb()
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
print("dequantize + recompose: %.3f ms  %.1f GB/s, max error %.3g of the tolerance" %
      (ms, u.nbytes / ms / 1e6, float((out - d).abs().max()) / (1e-3 * nrm)))
h.profile(True)
for _ in range(3): fb()
torch.cuda.synchronize()
for k, v in sorted(h.profile_read(reset=True).items(), key=lambda kv: -kv[1][0]):
    print("  back: %-16s %8.1f us/step  %4d launches/step" % (k, v[0] / 3 * 1e3, v[1] // 3))
