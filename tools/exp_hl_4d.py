"""configs[3] slab (8 x 512^3 f32) through mgh_compress / mgh_decompress and the low-level
dequantize+recompose: per-call times. Dev tool."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from mgard_amd import highlevel as hl
from tests.util import smooth_field
shape = (8, 512, 512, 512)
base = smooth_field(shape[1:], np.float32)
u = np.stack([base * np.float32(1.0 + 0.002 * t) + np.float32(1e-4 * t) for t in range(shape[0])])
d = torch.from_numpy(u).cuda()
nrm = float(np.max(np.abs(u)))
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf = hl.compress(d, 1e-3, float('inf'), mgard_amd.REL)
    torch.cuda.synchronize(); print("compress ms", (time.perf_counter() - t0) * 1e3, "ratio", u.nbytes / buf.numel())
out = torch.empty_like(d)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = hl.decompress(buf, out=out)
    torch.cuda.synchronize(); print("decompress ms", (time.perf_counter() - t0) * 1e3, "err", float((v - d).abs().max()), "<=", 1e-3 * nrm)
del buf, v, out
h = mgard_amd.Hierarchy(shape, np.float32)
q, oi, ov, cnt, n1 = h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), outlier_cap=d.numel() // 8)
out = torch.empty_like(d)
for i in range(4):
    qq = q.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.dequantize_recompose(qq, mgard_amd.REL, 1e-3, float('inf'), n1, outlier_idx=oi[:cnt], outlier_val=ov[:cnt], out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("dequantize+recompose ms", dt * 1e3, "GB/s out", u.nbytes / dt / 1e9)
h.profile(True)
h.dequantize_recompose(q.clone(), mgard_amd.REL, 1e-3, float('inf'), n1, outlier_idx=oi[:cnt], outlier_val=ov[:cnt], out=out)
print({k: round(v[0], 3) for k, v in h.profile_read().items()})
