"""Per-kernel times of dequantize + recompose at 512^3 f32: int64 input (the reference's interface)
and 16-bit symbols with the symbol width chosen per level (what mgh_decompress runs). Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from bench import gpu_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
d = gpu_field(torch, (n, n, n), torch.float32, dev)
h = mg.Hierarchy((n, n, n), np.float32)
q, oi, ov, cnt, nrm = h.decompose_quantize(d, mg.REL, 1e-3, float("inf"))
sym, si, sv, scnt, n2 = h.decompose_quantize_sym16(d, mg.REL, 1e-3, float("inf"))
out = torch.empty_like(d)
def t(f, k=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
qq = q.clone()
f64 = lambda: h.dequantize_recompose(qq, mg.REL, 1e-3, float("inf"), nrm, outlier_idx=oi, outlier_val=ov, out=out)
f16 = lambda: h.dequantize_recompose_sym16(sym, mg.REL, 1e-3, float("inf"), nrm, outlier_idx=si, outlier_val=sv, out=out)
for name, f in (("int64", f64), ("sym16 (per-level width)", f16)):
    print("%s: %.3f ms" % (name, t(f)))
    h.profile(True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    for k, v in sorted(h.profile_read(reset=True).items(), key=lambda kv: -kv[1][0]):
        print("   %-20s %7.1f us/step %3d launches" % (k, v[0] / 3 * 1e3, v[1] // 3))
    h.profile(False)
