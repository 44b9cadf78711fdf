"""Calibration: what streaming bandwidth do simple torch ops reach on this box for arrays of the
benchmark's size (read-only, copy, widening write)?"""
import time, torch
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
for N in (512**3, 1024**3):
    x=torch.randn(N,device='cuda'); y=torch.empty_like(x); q=torch.empty(N,dtype=torch.int64,device='cuda')
    r=t(lambda: y.copy_(x)); print(N,"copy f32   %.3f ms  %.2f TB/s"%(r*1e3, 8*N/r/1e12))
    r=t(lambda: torch.amax(x)); print(N,"amax       %.3f ms  %.2f TB/s"%(r*1e3, 4*N/r/1e12))
    r=t(lambda: x.abs().max()); print(N,"abs.max    %.3f ms"%(r*1e3))
    r=t(lambda: q.copy_(x)); print(N,"f32->i64   %.3f ms  %.2f TB/s"%(r*1e3, 12*N/r/1e12))
    r=t(lambda: q.zero_()); print(N,"memset i64 %.3f ms  %.2f TB/s"%(r*1e3, 8*N/r/1e12))
    del x,y,q
