import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field, nonuniform_coords
shape=(512,512,512)
for dt, coords, s, name in ((np.float64, nonuniform_coords(shape, np.float64), 0.0, "cfg3 f64 nonuniform s=0"),
                            (np.float64, None, float('inf'), "f64 uniform s=inf"),
                            (np.float32, None, 0.0, "f32 uniform s=0")):
    u=smooth_field(shape,dt); d=torch.from_numpy(u).cuda()
    h=mgard_amd.Hierarchy(shape,dt,coords=coords)
    cap=u.size//8
    bufs=(torch.empty(shape,dtype=torch.int64,device='cuda'),torch.zeros(1,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'))
    def step(): h.decompose_quantize(d, mgard_amd.REL, 1e-3, s, 0.0, bufs=bufs, want_norm=False)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/10
    h.profile(True)
    for _ in range(3): step()
    torch.cuda.synchronize(); p=h.profile_read(); h.profile(False)
    print(name, "ms/step %.3f"%(t*1e3), "GB/s in %.1f"%(u.nbytes/t/1e9), "outliers", int(bufs[1].item()), {k: round(v[0]/3*1000) for k,v in p.items()})
    h.close(); del d, bufs
