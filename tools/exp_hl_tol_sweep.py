"""mgh_compress / mgh_decompress against the tolerance (512^3 f32 bench field, device-resident): time,
ratio and error as more and more values leave the dictionary. Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from bench import gpu_field
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = gpu_field(torch, (n, n, n), torch.float32, dev)
nrm = float(d.abs().max())
out = torch.empty_like(d)
for tol in (1e-3, 1e-4, 3e-5, 1e-5, 1e-6):
    s = hl.compress(d, tol, float("inf"), mg.REL)
    for _ in range(2): s = hl.compress(d, tol, float("inf"), mg.REL)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): s = hl.compress(d, tol, float("inf"), mg.REL)
    torch.cuda.synchronize(); c = (time.perf_counter() - t0) / 5 * 1e3
    for _ in range(2): hl.decompress(s, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): hl.decompress(s, out=out)
    torch.cuda.synchronize(); dms = (time.perf_counter() - t0) / 5 * 1e3
    err = float((out - d).abs().max())
    print("tol %.0e: compress %.3f ms, decompress %.3f ms, ratio %.2f, error %.3g <= %.3g: %s" %
          (tol, c, dms, d.numel() * 4 / s.numel(), err, tol * nrm, err <= tol * nrm))
