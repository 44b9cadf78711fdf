"""mgh_compress / mgh_decompress on a device-resident 512^3 volume into a pre-allocated buffer (what
bench.py's end_to_end leg times)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
u = smooth_field((512, 512, 512), np.float32); ud = torch.from_numpy(u).cuda()
obuf = torch.empty(u.nbytes, dtype=torch.uint8, device='cuda')
out = torch.empty_like(ud)
for rep in range(3):
    for _ in range(2): s = hl.compress(ud, 1e-3, np.inf, mg.REL, out=obuf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): s = hl.compress(ud, 1e-3, np.inf, mg.REL, out=obuf)
    torch.cuda.synchronize(); c = (time.perf_counter() - t0) / 10 * 1e3
    hl.decompress(s, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hl.decompress(s, out=out)
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 10 * 1e3
    print("compress %.3f ms  decompress %.3f ms  ratio %.3f" % (c, d, u.nbytes / s.numel()))
