"""Per-shape step time and per-kernel microseconds (to attribute time to the top level:
compare n^3 with (n/2+1)^3)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
dt = np.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else np.float32
for n in [int(x) for x in sys.argv[1].split(",")]:
    shape=(n,n,n)
    u=smooth_field(shape,dt); d=torch.from_numpy(u).cuda()
    h=mgard_amd.Hierarchy(shape,dt)
    cap=max(u.size//8, 1024)
    bufs=(torch.empty(shape,dtype=torch.int64,device='cuda'),torch.zeros(1,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'))
    def step(): h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/20
    h.profile(True)
    for _ in range(5): step()
    torch.cuda.synchronize(); p=h.profile_read(); h.profile(False)
    print(n, "ms/step %.3f"%(t*1e3), {k: (round(v[0]/5*1000), v[1]//5) for k,v in p.items()})
    h.close(); del d, bufs
