"""configs[3] as ONE volume (nt x 512^3 f32, Variable {8} x nt/8 on dim 0) through mgh_compress /
mgh_decompress with the two-lane subdomain pipeline on and off (MGH_HL_PIPELINE), alternating.
Also 512^3 / 1024^3 single-subdomain calls for reference. Dev tool.
usage: python tools/exp_hl_pipeline.py [nt=64] [reps=3] [--once: one pipelined compress + decompress only, for rocprofv3]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd
from mgard_amd import highlevel as hl
from bench import gpu_field
args = [a for a in sys.argv[1:] if not a.startswith("--")]
nt = int(args[0]) if args else 64
reps = int(args[1]) if len(args) > 1 else 3
once = "--once" in sys.argv
dev = torch.device("cuda:0")
base = gpu_field(torch, (512, 512, 512), torch.float32, dev)
vol = torch.empty((nt, 512, 512, 512), dtype=torch.float32, device=dev)
for t in range(nt):
    vol[t] = base * (1.0 + 0.002 * t) + 1e-4 * t
del base
cfg = hl.Config(domain_decomposition=hl.DD_VARIABLE, domain_decomposition_dim=0,
                domain_decomposition_sizes=[8] * (nt // 8))
obuf = torch.empty(vol.numel() * 2, dtype=torch.uint8, device=dev)
back = torch.empty_like(vol)
nrm = float(vol.abs().max().item())


def run(mode, label=True):
    os.environ["MGH_HL_PIPELINE"] = mode
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = hl.compress(vol, 1e-3, float("inf"), mgard_amd.REL, config=cfg, out=obuf)
    torch.cuda.synchronize(); c = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    hl.decompress(s, out=back, config=cfg)
    torch.cuda.synchronize(); x = (time.perf_counter() - t0) * 1e3
    if label:
        print("pipeline=%s  compress %.2f ms  decompress %.2f ms  bytes %d" % (mode, c, x, s.numel()), flush=True)
    return s


if once:
    m = "0" if "--seq" in sys.argv else "1"
    run(m, False)
    run(m)
    sys.exit(0)
run("1", False); run("0", False)
for _ in range(reps):
    run("1"); s = run("0")
err = max(float((back[t] - vol[t]).abs().max().item()) for t in range(nt))
print("error %.3e <= %.3e: %s" % (err, 1e-3 * nrm, err <= 1e-3 * nrm))
