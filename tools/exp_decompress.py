import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
n=int(sys.argv[1]) if len(sys.argv)>1 else 512
shape=(n,n,n)
u=smooth_field(shape,np.float32); d=torch.from_numpy(u).cuda()
h=mgard_amd.Hierarchy(shape,np.float32)
q,oi,ov,cnt,nrm=h.decompose_quantize(d, mgard_amd.REL, 1e-3, float('inf'), outlier_cap=u.size//8)
out=torch.empty_like(d)
q2=q.clone()
for _ in range(3): h.dequantize_recompose(q2, mgard_amd.REL, 1e-3, float('inf'), nrm, outlier_idx=oi, outlier_val=ov, out=out)
torch.cuda.synchronize(); t0=time.perf_counter()
K=10
for _ in range(K): h.dequantize_recompose(q2, mgard_amd.REL, 1e-3, float('inf'), nrm, outlier_idx=oi, outlier_val=ov, out=out)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/K
print("decompress ms", dt*1e3, "GB/s (output)", u.nbytes/dt/1e9, "max err", float((out-d).abs().max()), "tol*norm", 1e-3*nrm)
h.profile(True)
for _ in range(3): h.dequantize_recompose(q2, mgard_amd.REL, 1e-3, float('inf'), nrm, outlier_idx=oi, outlier_val=ov, out=out)
torch.cuda.synchronize(); print({k: round(v[0]/3*1000,1) for k,v in h.profile_read().items()})
