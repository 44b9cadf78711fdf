"""mgh_compress on the bench field, many batches of 10 calls: is a batch ever far off the others?
(one evidence run of round 5 had a 2.4 ms batch between 1.30 ms ones). Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from bench import gpu_field
dev = torch.device("cuda:0")
d = gpu_field(torch, (512, 512, 512), torch.float32, dev)
obuf = torch.empty(d.numel() * 4 + 1000000, dtype=torch.uint8, device=dev)
for _ in range(2): s = hl.compress(d, 1e-3, float("inf"), mg.REL, out=obuf)
torch.cuda.synchronize()
ts = []
for b in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    t0 = time.perf_counter()
    for _ in range(10): s = hl.compress(d, 1e-3, float("inf"), mg.REL, out=obuf)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 10 * 1e3)
ts = np.array(ts)
print("batches %d: min %.3f median %.3f max %.3f ms; batches above 1.2 x median: %s" %
      (len(ts), ts.min(), np.median(ts), ts.max(), [round(float(x), 3) for x in ts[ts > 1.2 * np.median(ts)]]))
# single calls, synchronised
one = []
for _ in range(300):
    t0 = time.perf_counter(); s = hl.compress(d, 1e-3, float("inf"), mg.REL, out=obuf); torch.cuda.synchronize()
    one.append((time.perf_counter() - t0) * 1e3)
one = np.array(one)
print("single calls: min %.3f median %.3f p99 %.3f max %.3f ms" % (one.min(), np.median(one), np.percentile(one, 99), one.max()))
