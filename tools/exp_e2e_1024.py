"""BASELINE.json configs[4]: 1024^3 float32 compress + decompress round trip through the container,
error-vs-tolerance check, end-to-end GB/s (device-resident in and out)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
n = 1024
u = smooth_field((n, n, n), np.float32)
ud = torch.from_numpy(u).cuda()
nrm = float(np.max(np.abs(u)))
del u
def t(f, k=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k, r
tc, buf = t(lambda: hl.compress(ud, 1e-3, np.inf, mg.REL))
td, v = t(lambda: hl.decompress(buf))
err = float((v - ud).abs().max().item())
nb = ud.numel() * 4
print("1024^3 f32 REL 1e-3: compress %.1f ms (%.1f GB/s), decompress %.1f ms (%.1f GB/s), ratio %.2f, "
      "Linf error %.3e <= %.3e: %s" % (tc * 1e3, nb / tc / 1e9, td * 1e3, nb / td / 1e9, nb / buf.numel(),
                                       err, 1e-3 * nrm, err <= 1e-3 * nrm))
h = mg.Hierarchy((n, n, n), np.float32)
cap = ud.numel() // 8
bufs = (torch.empty((n, n, n), dtype=torch.int64, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda'),
        torch.empty(cap, dtype=torch.int64, device='cuda'), torch.empty(cap, dtype=torch.int64, device='cuda'))
ts, _ = t(lambda: h.decompose_quantize(ud, mg.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False), 5)
print("1024^3 f32 decompose+quantize only: %.2f ms = %.1f GB/s (input)" % (ts * 1e3, nb / ts / 1e9))
