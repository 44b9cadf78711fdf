"""End-to-end high-level compress / decompress timing (container in, container out)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
u = smooth_field((n, n, n), np.float32)
ud = torch.from_numpy(u).cuda()
def t(f, k=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k, r
for name, cfg in (("huffman", hl.Config()), ("huffman+zstd", hl.Config(lossless=hl.HUFFMAN_ZSTD))):
    tc, buf = t(lambda: hl.compress(ud, 1e-3, np.inf, mg.REL, config=cfg))
    td, v = t(lambda: hl.decompress(buf))
    print("%-13s device: compress %.1f ms (%.2f GB/s)  decompress %.1f ms (%.2f GB/s)  CR %.2f" % (
        name, tc * 1e3, u.nbytes / tc / 1e9, td * 1e3, u.nbytes / td / 1e9, u.nbytes / buf.numel()))
tc, buf = t(lambda: hl.compress(u, 1e-3, np.inf, mg.REL), 2)
td, v = t(lambda: hl.decompress(buf), 2)
print("huffman       host  : compress %.1f ms (%.2f GB/s)  decompress %.1f ms (%.2f GB/s)" % (
    tc * 1e3, u.nbytes / tc / 1e9, td * 1e3, u.nbytes / td / 1e9))
for pin in (1, 0):
    cfg = hl.Config(auto_pin_host_buffers=pin)
    tc, buf = t(lambda: hl.compress(u, 1e-3, np.inf, mg.REL, config=cfg), 2)
    print("host compress, auto_pin_host_buffers=%d: %.1f ms" % (pin, tc * 1e3))
