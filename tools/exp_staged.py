"""The stage-by-stage low-level path (what a per-stage binding of mgard_x::Compressor uses) next to
the fused entry point, 512^3 f32."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from tests.util import smooth_field
shape = (512, 512, 512)
u = smooth_field(shape, np.float32); d = torch.from_numpy(u).cuda()
h = mg.Hierarchy(shape, np.float32)
coef = torch.empty_like(d)
def t(f, k=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
nrm = h.norm(d, float('inf'))
print("norm (host value)      %.3f ms" % t(lambda: h.norm(d, float('inf'))))
print("decompose              %.3f ms" % t(lambda: h.decompose(d, out=coef)))
print("quantize               %.3f ms" % t(lambda: h.quantize(coef, mg.REL, 1e-3, float('inf'), nrm, outlier_cap=u.size // 8)))
q, oi, ov, cnt = h.quantize(coef, mg.REL, 1e-3, float('inf'), nrm, outlier_cap=u.size // 8)
print("dequantize             %.3f ms" % t(lambda: h.dequantize(q.clone(), mg.REL, 1e-3, float('inf'), nrm, outlier_idx=oi, outlier_val=ov)))
c2 = h.dequantize(q.clone(), mg.REL, 1e-3, float('inf'), nrm, outlier_idx=oi, outlier_val=ov)
print("recompose              %.3f ms" % t(lambda: h.recompose(c2)))
