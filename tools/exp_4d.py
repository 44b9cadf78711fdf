import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
shape=tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv)>1 else (8,256,256,256)
# a slab of time steps: a 3-D field that evolves slowly along dim 0 (smooth_field oscillates along
# every dim, which with 8 samples along dim 0 makes half the nodes outliers)
if len(shape) == 4:
    base = smooth_field(shape[1:], np.float32)
    u = np.stack([base * np.float32(1.0 + 0.002 * t) + np.float32(1e-4 * t) for t in range(shape[0])])
else:
    u = smooth_field(shape, np.float32)
d=torch.from_numpy(u).cuda()
h=mgard_amd.Hierarchy(shape,np.float32)
print("shape",shape,"l_target",h.l_target)
cap=u.size//8
bufs=(torch.empty(shape,dtype=torch.int64,device='cuda'),torch.zeros(1,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'))
def step(): return h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=True)
step(); torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); t=(time.perf_counter()-t0)/3
h.profile(True); step(); torch.cuda.synchronize(); p=h.profile_read()
print("ms/step %.2f GB/s %.1f"%(t*1e3,u.nbytes/t/1e9), "outliers", int(bufs[1].item()), {k: (round(v[0],2), v[1]) for k,v in p.items()})
