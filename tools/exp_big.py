"""Large volumes through the high-level path, generated on the device (no host copy of the data):
error bound and throughput; 2048^3 needs the memory-driven domain decomposition."""
import sys, time, math, torch
sys.path.insert(0, '/root/repo')
import mgard_amd as mg
from mgard_amd import highlevel as hl
n = int(sys.argv[1])
x = torch.linspace(0, 1, n, device='cuda')
u = (torch.sin(2 * math.pi * 3 * x)[None, None, :] * torch.cos(2 * math.pi * 2 * x)[None, :, None]).expand(n, n, n).contiguous()
u += 0.5 * torch.sin(2 * math.pi * 5 * x)[:, None, None]
g = torch.Generator(device='cuda'); g.manual_seed(1)
for i in range(0, n, 64):   # noise in slabs (keeps the temporary small)
    u[i:i + 64] += 1e-3 * (2 * torch.rand(u[i:i + 64].shape, device='cuda', generator=g) - 1)
nrm = float(u.abs().max().item())
nb = u.numel() * 4
print("n", n, "GB", nb / 1e9, "free GB", torch.cuda.mem_get_info()[0] / 1e9, flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
buf = hl.compress(u, 1e-3, float('inf'), mg.REL)
torch.cuda.synchronize(); tc = time.perf_counter() - t0
m = hl.metadata_parse(bytes(buf[:4096].cpu().numpy()))
print("compress %.1f ms (%.1f GB/s) ratio %.2f decomposed %s dd_dim %d dd_size %d" % (
    tc * 1e3, nb / tc / 1e9, nb / buf.numel(), m["domain_decomposed"], m["dd_dim"], m["dd_size"]), flush=True)
outbuf = torch.empty(nb // 2, dtype=torch.uint8, device='cuda')
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf = hl.compress(u, 1e-3, float('inf'), mg.REL, out=outbuf)
    torch.cuda.synchronize(); tc = time.perf_counter() - t0
    print("compress (warm, pre-allocated output) %.1f ms (%.1f GB/s)" % (tc * 1e3, nb / tc / 1e9), flush=True)
t0 = time.perf_counter()
v = hl.decompress(buf)
torch.cuda.synchronize(); td = time.perf_counter() - t0
print("decompress (cold) %.1f ms" % (td * 1e3), flush=True)
del v
t0 = time.perf_counter()
v = hl.decompress(buf)
torch.cuda.synchronize(); td = time.perf_counter() - t0
err = 0.0
for i in range(0, n, 64):
    err = max(err, float((v[i:i + 64] - u[i:i + 64]).abs().max().item()))
print("decompress %.1f ms (%.1f GB/s) Linf error %.3e <= %.3e: %s" % (td * 1e3, nb / td / 1e9, err, 1e-3 * nrm, err <= 1e-3 * nrm))
