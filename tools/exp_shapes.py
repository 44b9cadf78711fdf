"""mgh_compress / mgh_decompress over shapes of every dimensionality (device-resident, 64-270 MB each):
where the paths beside the fused 3-D / 4-D one stand. Run from a checkout's root. Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
def t(f, k=3):
    for _ in range(2): r = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3, r
for shape, dt in (((1 << 24,), np.float32), ((4096, 4096), np.float32), ((8192, 8192), np.float32), ((256, 256, 256), np.float32),
                  ((256, 256, 256), np.float64), ((64, 64, 64, 64), np.float32), ((8, 8, 64, 64, 64), np.float32),
                  ((300, 311, 322), np.float32), ((2048, 2048, 17), np.float32), ((17, 2048, 2048), np.float32),
                  ((2048, 17, 2048), np.float32), ((100, 100, 6000), np.float32), ((6000, 100, 100), np.float32),
                  ((1 << 22, 9), np.float32), ((9, 1 << 22), np.float32), ((5, 5, 5, 5, 40000), np.float32),
                  ((16384, 4097), np.float64), ((3, 3, 1 << 22), np.float32), ((1 << 22, 3, 3), np.float32)):
    u = torch.from_numpy(smooth_field(shape, dt)).cuda()
    nb = u.numel() * u.element_size()
    try:
        c, s = t(lambda: hl.compress(u, 1e-3, np.inf, mg.REL))
        d, v = t(lambda: hl.decompress(s))
        print("%-22s %-8s compress %8.2f ms (%6.1f GB/s)  decompress %8.2f ms (%6.1f GB/s)  ratio %.2f" % (
            "x".join(map(str, shape)), np.dtype(dt).name, c, nb / c / 1e6, d, nb / d / 1e6, nb / s.numel()))
    except Exception as e:
        print(shape, "failed:", str(e)[:100])
