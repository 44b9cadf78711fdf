"""Does the step time depend on where the output array sits relative to the input (HBM channel
interleave)? One process, one input, the int64 output at different byte offsets inside one pool."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
n1 = int(os.environ.get("NSIDE", "512"))
shape = (n1, n1, n1)
N = n1 ** 3
u = smooth_field(shape, np.float32)
uoff = int(os.environ.get("UOFF", "0"))
upool = torch.empty(N + 4096, dtype=torch.float32, device='cuda')
d = upool[uoff // 4: uoff // 4 + N].view(shape)
d.copy_(torch.from_numpy(u))
h = mgard_amd.Hierarchy(shape, np.float32)
cap = N // 16
pool = torch.empty(N + (64 << 20) // 8, dtype=torch.int64, device='cuda')
cnt = torch.zeros(1, dtype=torch.int64, device='cuda')
oidx = torch.empty(cap, dtype=torch.int64, device='cuda'); oval = torch.empty(cap, dtype=torch.int64, device='cuda')
print("input %x pool %x" % (d.data_ptr(), pool.data_ptr()))
def run(q, n=30):
    bufs = (q, cnt, oidx, oval)
    for _ in range(3): h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
offs = [int(x) for x in os.environ.get("OFFS", "0,64,128,192,256,320,384,512,768,1024,1280,2048,2304,4096,4352,8192").split(",")]
for rep in range(2):
    for off in offs:
        q = pool[off // 8: off // 8 + N].view(shape)
        print("offset %9d  %.4f ms" % (off, run(q)))
