"""mgh_compress / mgh_decompress with HOST buffers (numpy in, numpy out): whole array and block-decomposed.
Run from a checkout's root. Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
def t(f, k=5):
    for _ in range(2): r = f()
    t0 = time.perf_counter()
    for _ in range(k): r = f()
    return (time.perf_counter() - t0) / k * 1e3, r
for n, cfg, label in ((257, None, "257^3"), (257, hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=129), "257^3 in 8 blocks of 129"),
                      (512, None, "512^3"), (512, hl.Config(domain_decomposition=hl.DD_BLOCK, block_size=256), "512^3 in 8 blocks of 256"),
                      (512, hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=0, domain_decomposition_sizes=[128] * 4), "512^3 in 4 slabs")):
    u = smooth_field((n, n, n), np.float32)
    c, s = t(lambda: hl.compress(u, 1e-3, np.inf, mg.REL, config=cfg))
    d, v = t(lambda: hl.decompress(s, config=cfg))
    print("%-28s compress %7.2f ms  decompress %7.2f ms  (%d bytes)" % (label, c, d, len(s)))
