"""Does an idle gap in front of a step change how fast its kernels run (power / clock management)?
Per-kernel times of the step with torch.cuda._sleep(cycles) between the steps. Dev tool.
usage: python tools/exp_idle.py [n]"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from bench import gpu_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
dev = torch.device("cuda", 0)
d = gpu_field(torch, shape, torch.float32, dev)
h = mgard_amd.Hierarchy(shape, np.float32)
cap = d.numel() // 16
bufs = (torch.empty(shape, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
        torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int64, device=dev))
def step():
    h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
for _ in range(3):
    step()
for gap_us in (0, 100, 300, 1000, 3000, 0):
    cycles = int(gap_us * 100)   # _sleep counts ~100 MHz wall-clock ticks? (calibrated below)
    torch.cuda.synchronize()
    h.profile(True)
    K = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        if cycles:
            torch.cuda._sleep(cycles)
        step()
    e1.record()
    torch.cuda.synchronize()
    p = h.profile_read(reset=True)
    h.profile(False)
    tot = e0.elapsed_time(e1) / K
    ksum = sum(v[0] for v in p.values()) / K
    print("gap %5d us: loop %.3f ms/iter, kernels %.3f ms, pass %.1f us, absmax %.1f us, solves %.1f us" % (
        gap_us, tot, ksum, p["level_fused_q"][0] / K * 1e3, p["absmax"][0] / K * 1e3,
        (p["ipk_f"][0] + p["ipk_c"][0] + p["ipk_r"][0]) / K * 1e3))
