"""mgh_compress / mgh_decompress on small and on block-decomposed inputs: per-call overhead of the
high-level path. Run from a checkout's root: python tools/exp_small_calls.py. Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
def t(f, k=20):
    for _ in range(3): r = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3, r
for n, cfg, label in ((65, None, "65^3"), (129, None, "129^3"), (257, None, "257^3"),
                      (257, hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=129), "257^3 in 8 blocks of 129"),
                      (257, hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=65), "257^3 in 64 blocks of 65")):
    u = torch.from_numpy(smooth_field((n, n, n), np.float32)).cuda()
    c, s = t(lambda: hl.compress(u, 1e-3, np.inf, mg.REL, config=cfg))
    d, v = t(lambda: hl.decompress(s, config=cfg))
    print("%-28s compress %7.3f ms  decompress %7.3f ms" % (label, c, d))
