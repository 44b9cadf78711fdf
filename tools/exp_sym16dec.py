"""mgh_decompress with int64 vs 16-bit symbols between decoder and dequantizer (MGH_SYM16_DECODE) on
fields with few and with many outliers (run once per setting: the switch is read once per process)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
for n, tol in ((256, 1e-3), (256, 1e-5), (256, 1e-7), (512, 1e-3), (512, 1e-5)):
    u = smooth_field((n, n, n), np.float32); ud = torch.from_numpy(u).cuda()
    s = hl.compress(ud, tol, np.inf, mg.REL)
    out = torch.empty_like(ud)
    for _ in range(2): hl.decompress(s, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hl.decompress(s, out=out)
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 10 * 1e3
    print("n %d tol %.0e ratio %.2f decompress %.3f ms" % (n, tol, u.nbytes / s.numel(), d))
