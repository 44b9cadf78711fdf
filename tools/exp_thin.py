"""The low-level step (norm + decompose + quantize into 16-bit symbols) on shapes with short fast
dimensions, on a field inside the dictionary: which property of a shape costs what. Dev tool.
  python tools/exp_thin.py "16395,39,39:f64" "8,16395,39,39:f64" "512,512,512:f32:0" ...   (shape:type[:s])"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
def field(shape, dt):
    ax = np.meshgrid(*[np.arange(n, dtype=np.float64) / max(n - 1, 1) for n in shape], indexing="ij", sparse=True)
    g = sum((1.0 if shape[k] >= 32 else 0.01) * np.sin(2 * np.pi * a + 0.3 * k) for k, a in enumerate(ax))
    g = g + 1e-3 * np.random.default_rng(1).uniform(-1, 1, size=shape)
    return np.ascontiguousarray(g.astype(dt))
for spec in sys.argv[1:]:
    parts = spec.split(":")
    sh, ty = parts[0], parts[1]
    S = float(parts[2]) if len(parts) > 2 else float("inf")   # smoothness parameter (default: L-infinity)
    shape = tuple(int(x) for x in sh.split(","))
    dt = np.float64 if ty == "f64" else np.float32
    u = torch.from_numpy(field(shape, dt)).cuda()
    h = mg.Hierarchy(shape, dt)
    f = (lambda: h.decompose_quantize_sym16(u, mg.REL, 1e-3, S)) if h.sym16_supported() else \
        (lambda: h.decompose_quantize(u, mg.REL, 1e-3, S))
    for _ in range(2): r = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    nb = u.numel() * u.element_size()
    print("%-24s %s s=%g %8.3f ms  %7.1f GB/s  outliers %d  L=%d" % (sh, ty, S, ms, nb / ms / 1e6, int(r[3]), h.l_target))
    h.profile(True)
    f(); torch.cuda.synchronize()
    for k, v in sorted(h.profile_read(reset=True).items(), key=lambda kv: -kv[1][0])[:4]:
        print("      %-22s %8.1f us  %3d launches" % (k, v[0] * 1e3, v[1]))
    if h.sym16_supported():
        out = torch.empty_like(u)
        n = int(r[3])
        fb = lambda: h.dequantize_recompose_sym16(r[0], mg.REL, 1e-3, S, r[4], outlier_idx=r[1][:n], outlier_val=r[2][:n], out=out)
        for _ in range(2): fb()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fb()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        print("      back: %8.3f ms  %7.1f GB/s  (max error %.2g of tol * norm)" % (ms, nb / ms / 1e6, float((out - u).abs().max()) / (1e-3 * r[4])))
        h.profile(True)
        fb(); torch.cuda.synchronize()
        for k, v in sorted(h.profile_read(reset=True).items(), key=lambda kv: -kv[1][0])[:3]:
            print("      back  %-16s %8.1f us  %3d launches" % (k, v[0] * 1e3, v[1]))
    h.close()
