"""512^3 f32 dequantize+recompose: per-kernel times. Dev tool."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
shape = (512, 512, 512)
u = smooth_field(shape, np.float32)
d = torch.from_numpy(u).cuda()
h = mgard_amd.Hierarchy(shape, np.float32)
q, oi, ov, cnt, n1 = h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), outlier_cap=d.numel() // 8)
out = torch.empty_like(d)
qq = q.clone()
for i in range(3):
    h.dequantize_recompose(qq, mgard_amd.REL, 1e-3, float('inf'), n1, outlier_idx=oi[:cnt], outlier_val=ov[:cnt], out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20):
    h.dequantize_recompose(qq, mgard_amd.REL, 1e-3, float('inf'), n1, outlier_idx=oi[:cnt], outlier_val=ov[:cnt], out=out)
torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) / 20 * 1e3, "err", float((out - d).abs().max()))
h.profile(True)
for i in range(5):
    h.dequantize_recompose(qq, mgard_amd.REL, 1e-3, float('inf'), n1, outlier_idx=oi[:cnt], outlier_val=ov[:cnt], out=out)
print({k: round(v[0] / 5 * 1000) for k, v in h.profile_read().items()})
