"""Step time of decompose + quantize against the tolerance (512^3 f32 bench field): how the step
behaves when more and more values leave the dictionary (MGH_OUTLIER_AGG=0: always the per-wave slot requests). Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from bench import gpu_field
dev = torch.device("cuda:0")
# argument: the side of a cube (512) or a shape (8,512,512,512)
arg = sys.argv[1] if len(sys.argv) > 1 else "512"
shape = tuple(int(x) for x in arg.split(",")) if "," in arg else (int(arg),) * 3
d = gpu_field(torch, shape, torch.float32, dev)
h = mg.Hierarchy(shape, np.float32)
N = d.numel()
q = torch.empty(shape, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
oi = torch.empty(N, dtype=torch.int64, device=dev); ov = torch.empty(N, dtype=torch.int64, device=dev)
for tol in (1e-3, 1e-4, 3e-5, 1e-5, 1e-6):
    f = lambda: h.decompose_quantize(d, mg.REL, tol, float("inf"), 0.0, bufs=(q, cnt, oi, ov), want_norm=False)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print("tol %.0e: %.3f ms per step, %d outliers (%.1f %% of the values)" % (tol, ms, int(cnt.item()), 100.0 * int(cnt.item()) / N))
