import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
shape=(512,512,512)
u=smooth_field(shape,np.float32); d=torch.from_numpy(u).cuda()
h=mgard_amd.Hierarchy(shape,np.float32)
cap=u.size//16
bufs=(torch.empty(shape,dtype=torch.int64,device='cuda'),torch.zeros(1,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'))
def step(): h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False)
for _ in range(3): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter()-t0)/20*1e3)
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): step()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
torch.cuda.synchronize()
ref=bufs[0].clone(); bufs[0].zero_()
g.replay(); torch.cuda.synchronize()
print("graph result equal:", torch.equal(ref,bufs[0]))
t0=time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter()-t0)/20*1e3)
