"""Why is the Huffman decoder slow on a shape? Symbol statistics of the subdomain (16-bit symbols of
decompose + quantize) and decode time through the lossless stage alone. Dev tool.
  python tools/exp_decode_diag.py 100,100,6000"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
shape = tuple(int(x) for x in sys.argv[1].split(","))
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
u = torch.from_numpy(smooth_field(shape, np.float32)).cuda()
h = mg.Hierarchy(shape, np.float32)
q, oi, ov, n, nrm = h.decompose_quantize(u, mg.REL, tol, np.inf)
qq = q.flatten()
cnt = torch.bincount(qq.clamp(0, 8191).to(torch.int64), minlength=8192).cpu().numpy().astype(np.float64)
p = cnt / cnt.sum()
used = int((cnt > 0).sum())
ent = float(-(p[p > 0] * np.log2(p[p > 0])).sum())
top = np.sort(p)[::-1][:6]
print("shape %s: %d symbols, %d outliers (%.1f %%), %d dictionary entries used, entropy %.2f bits, top shares %s" %
      (shape, qq.numel(), n, 100.0 * n / qq.numel(), used, ent, np.round(top, 4)))
L = hl.Lossless()
rec = L.compress(qq, 8192, 20480, outlier_idx=oi[:n].contiguous(), outlier_val=ov[:n].contiguous())
rd = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
for _ in range(2): back = L.decompress(rd, qq.numel())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): back = L.decompress(rd, qq.numel())
torch.cuda.synchronize()
print("  record %d bytes (%.2f bits per symbol incl. outliers), decode %.2f ms" % (len(rec), 8.0 * len(rec) / qq.numel(), (time.perf_counter() - t0) / 3 * 1e3))
# the same symbols in another order (sorted: long runs of one symbol) -- is it the order?
qs, _ = torch.sort(qq)
rec2 = L.compress(qs, 8192, 20480)
rd2 = torch.frombuffer(bytearray(rec2), dtype=torch.uint8).cuda()
for _ in range(2): back = L.decompress(rd2, qq.numel())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): back = L.decompress(rd2, qq.numel())
torch.cuda.synchronize()
print("  sorted symbols: decode %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
