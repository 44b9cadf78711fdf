"""Lossless stage alone: encode / decode time versus the entropy of the symbols."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from mgard_amd import highlevel as hl
n = 64 * 1024 * 1024
ctx = hl.Lossless()
for width in (2, 20, 200, 800):
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    q = (torch.randn(n, device='cuda', generator=g) * width).round().to(torch.int64) + 4096
    q.clamp_(0, 8191)
    rec = ctx.compress(q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): rec = ctx.compress(q)
    torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 3
    back, _, _ = ctx.decompress(rec, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): back, _, _ = ctx.decompress(rec, n)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 3
    assert torch.equal(back, q)
    print("width %4d: %.2f bits/sym  compress %.1f ms (incl. D2H of the record)  decompress %.1f ms (incl. H2D)" % (
        width, len(rec) * 8 / n, tc * 1e3, td * 1e3))
