import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
shape=(512,512,512)
u=smooth_field(shape,np.float32); d=torch.from_numpy(u).cuda()
h=mgard_amd.Hierarchy(shape,np.float32)
cap=u.size//16
bufs=(torch.empty(shape,dtype=torch.int64,device='cuda'),torch.zeros(1,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'),torch.empty(cap,dtype=torch.int64,device='cuda'))
for dict_size in (8192, 1<<30):
    for _ in range(3): h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False, dict_size=dict_size)
    torch.cuda.synchronize(); h.profile(True)
    for _ in range(5): h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), 0.0, bufs=bufs, want_norm=False, dict_size=dict_size)
    torch.cuda.synchronize(); p=h.profile_read(); h.profile(False)
    print(dict_size, int(bufs[1].item()), {k: round(v[0]/5*1000,1) for k,v in p.items()})
