"""1-D and 2-D inputs (one-thread-per-element kernels): decompose+quantize throughput."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
for shape in [(8192, 8192), (4097, 4097), (1 << 26,)]:
    n = int(np.prod(shape))
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    d = torch.rand(shape, device='cuda', generator=g)
    h = mg.Hierarchy(shape, np.float32)
    cap = n // 4
    bufs = (torch.empty(shape, dtype=torch.int64, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda'),
            torch.empty(cap, dtype=torch.int64, device='cuda'), torch.empty(cap, dtype=torch.int64, device='cuda'))
    def step(): h.decompose_quantize(d, mg.REL, 1e-2, float('inf'), 0.0, bufs=bufs, want_norm=True)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    h.profile(True); step(); torch.cuda.synchronize(); p = h.profile_read()
    print(shape, "L", h.l_target, "ms/step %.2f  GB/s %.1f" % (t * 1e3, n * 4 / t / 1e9),
          {k: (round(v[0], 2), v[1]) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])[:6]})
    h.close()
