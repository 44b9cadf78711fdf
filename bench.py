#!/usr/bin/env python3
"""bench.py -- compress GB/s (input) of the MGARD-X hot path on MI355X.

A "step" is one pass of [norm +] multilevel decomposition + level-wise quantization
(what mgard_x::Compressor::Compress runs before the lossless stage,
reference include/mgard-x/CompressionLowLevel/Compressor.hpp:216-218) over one synthetic
512^3 float32 volume that is already resident in HBM, REL L-inf tolerance 1e-3
(BASELINE.json configs[1]).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling. Every rank owns one
512^3 subdomain of a (N*512) x 512 x 512 volume split along the slowest dimension, exactly like
the reference's domain decomposition (independent subdomains, no halo;
include/mgard-x/DomainDecomposer/DomainDecomposer.hpp:260-303). The only data-path exchange is
the scalar all-reduce (MAX over RCCL) of the subdomain norms that a REL bound needs
(include/mgard-x/CompressionHighLevel/ErrorToleranceCalculator.hpp:69-89,134-155).

`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment is a LAUNCHER: before
anything touches a GPU it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child process (one rank per GPU over RCCL), relays rank 0's JSON line and adds a second
leg from another child -- `mgh_compress_multi`, one process driving N devices -- as
`native_multi`. Under `torch.distributed.run` (WORLD_SIZE set) it is a rank.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TOL = 1e-3
SHAPE = (512, 512, 512)


def algorithmic_bytes_per_step(h, esz):
    """Algorithmic HBM bytes per step for every kernel name the library launches (DESIGN.md,
    "Kernels"): what each kernel must read and write once, summed over the levels it runs on."""
    L = h.l_target
    out = {}

    def add(name, b):
        out[name] = out.get(name, 0) + b

    def vol(s):
        v = 1
        for x in s:
            v *= x
        return v

    N = vol(h.shape)
    add("absmax", N * esz)
    add("quantize", N * (esz + 8))
    if len(h.shape) == 4:
        # D = 4 (decompose_fused4): per level the even t-slices run the 3-D pass of the slice, the
        # odd ones read their own slice once (the neighbours' even planes are re-reads), write a
        # coefficient for every node and the slice's load vector
        for l in range(L, 0, -1):
            n, m = h.level_shape(l), h.level_shape(l - 1)
            n3, m3 = vol(n[1:]), vol(m[1:])
            add("level4_even", m[0] * (n3 * esz + (n3 - m3) * 8 + 2 * m3 * esz))
            add("level4_odd", (n[0] - m[0]) * (n3 * esz + n3 * 8 + m3 * esz))
            add("tsweep", ((2 * m[0] - 1) + m[0]) * m3 * esz)
            add("ipk_f", 2 * vol(m) * esz)
            add("ipk_c", 2 * vol(m) * esz)
            add("ipk_r", 2 * vol(m) * esz)
            add("ipk_t", 3 * vol(m) * esz)  # the t solve reads the correction and applies it to the coarse nodes
        return out
    for l in range(L, 0, -1):
        n = h.level_shape(l)
        m = h.level_shape(l - 1)
        n = (1,) * (3 - len(n)) + tuple(n)
        m = (1,) * (3 - len(m)) + tuple(m)
        add("gpk_reo", 2 * vol(n) * esz)
        add("lpk_f", (n[0] * n[1] * n[2] + n[0] * n[1] * m[2]) * esz)
        add("lpk_c", (n[0] * n[1] * m[2] + n[0] * m[1] * m[2]) * esz)
        add("lpk_r", (n[0] * m[1] * m[2] + vol(m)) * esz)
        add("ipk_f", 2 * vol(m) * esz)
        add("ipk_c", 2 * vol(m) * esz)
        add("ipk_r", 4 * vol(m) * esz)  # + read/modify/write of the coarse nodes
        # fused kernels (kernels_fast.hpp) -- see DESIGN.md
        # fused level kernel: read the fine nodes once, write the quantized coefficients
        # (int64), the coarse nodes and the load vector
        # (capi.hip: levels with >= 2048 tiles of 8x32x16 coarse nodes use the long-march
        # variant "level_fused_q", smaller ones "level_fused_q_small")
        # march classes of capi.hip (level_class): long marches "level_fused_q", mid-size
        # "level_fused_q_small", few tiles: the box kernel "level_box_q" (kernels_box.hpp)
        tiles = -(-m[2] // 32) * -(-m[1] // 8)
        big = tiles * -(-m[0] // 16) >= 2048
        mid = tiles * -(-m[0] // 4) >= 256
        add("level_fused_q" if big else ("level_fused_q_small" if mid else "level_box_q"),
            vol(n) * esz + (vol(n) - vol(m)) * 8 + 2 * vol(m) * esz)
    add("copy_box", 2 * vol((1,) * (3 - len(h.shape)) + tuple(h.level_shape(0))) * esz)
    return out


TRAFFIC_FILES = {"512f32": "traffic_512cube_f32.json", "1024f32": "traffic_1024cube_f32.json",
                 "512f64nu": "traffic_512cube_f64nu.json", "4d": "traffic_4d_slab_f32.json"}


def pmc_traffic(config_name, kernel):
    """(bytes per launch or None, note) of `kernel` from the committed rocprofv3 PMC passes of this
    configuration (profiles/traffic_*.json, tools/make_traffic.py). A file taken on other kernel
    sources than the ones this run loads is refused."""
    tname = TRAFFIC_FILES.get(config_name)
    tpath = os.path.join(ROOT, "profiles", tname or "none")
    if not tname or not os.path.exists(tpath):
        return None, "no PMC traffic file for this configuration"
    tj = json.load(open(tpath))
    if tj.get("source_hash") != source_hash():
        return None, "profiles/%s was taken on other kernel sources (hash %s, now %s): refused" % (
            tname, tj.get("source_hash"), source_hash())
    t = tj.get(kernel)
    if not t:
        return None, "profiles/%s has no entry for %s" % (tname, kernel)
    return (int((t["fetch_kib"] * t["read_correction"] + t["write_kib"]) * 1024),
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on these kernel sources, per launch")


def _oracle_step(o, oracle, u, tol, s):
    import numpy as np
    t0 = time.perf_counter()
    nrm = oracle.norm(u, u.dtype.type(s))
    c = o.decompose(u)
    q, oi, ov, n = o.quantize(c, oracle.REL, u.dtype.type(tol), u.dtype.type(s), u.dtype.type(nrm))
    return time.perf_counter() - t0, q


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; None when the file does not say."""
    try:
        seen, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(u, tol, s, coords=None):
    """The CPU oracle (a port of the reference algorithm, OpenMP over pencils) timed on the host
    of the GPU box, beside the GPU numbers of the same run. Reported baseline only.
    (i) all host cores on the whole workload: one untimed warm-up, then the MEDIAN of 5 steps;
    (ii) one thread (continuity with the reference's SERIAL backend) on a bounded sample --
    the leading 128 planes of the same volume -- median of 3."""
    import numpy as np
    import oracle
    cores = oracle.num_threads()
    o = oracle.Hierarchy(u.shape, u.dtype, coords=coords)
    _oracle_step(o, oracle, u, tol, s)
    ts = []
    q = None
    for _ in range(5):
        dt, q = _oracle_step(o, oracle, u, tol, s)
        ts.append(dt)
    med = sorted(ts)[len(ts) // 2]
    out = {"value": u.nbytes / med / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
           "nproc": os.cpu_count(), "physical_cores": physical_cores(),
           "affinity_cpus": len(os.sched_getaffinity(0)),
           "threads_note": "cores = OpenMP threads used = omp_get_max_threads() of this process, which "
                           "importing torch sets to one per PHYSICAL core (128 of the 256 logical CPUs on "
                           "the MI355X hosts seen so far); nproc = logical CPUs of the host, "
                           "physical_cores = distinct (socket, core) pairs of /proc/cpuinfo, "
                           "affinity_cpus = CPUs this process may run on",
           "sample": "the same %s %s workload (norm+decompose+quantize): 1 warm-up, median of 5 "
                     "steps = %.2f s (min %.2f, max %.2f)"
                     % ("x".join(map(str, u.shape)), u.dtype.name, med, min(ts), max(ts))}
    try:
        sub = np.ascontiguousarray(u[:min(128, u.shape[0])])
        sc = None if coords is None else [coords[0][:sub.shape[0]]] + list(coords[1:])
        oracle.set_num_threads(1)
        o1 = oracle.Hierarchy(sub.shape, sub.dtype, coords=sc)
        _oracle_step(o1, oracle, sub, tol, s)
        t1 = sorted(_oracle_step(o1, oracle, sub, tol, s)[0] for _ in range(3))[1]
        out["serial"] = {"value": sub.nbytes / t1 / 1e9, "unit": "GB/s", "cores": 1,
                         "sample": "leading %s block of the volume, 1 warm-up, median of 3 = %.2f s"
                                   % ("x".join(map(str, sub.shape)), t1)}
    finally:
        oracle.set_num_threads(cores)
    return out, q


def source_hash():
    """Hash of the kernel sources of the low-level library: capi.hip and the kernels_*.hpp /
    hierarchy.hpp it includes, directly or not -- the code of the measured step. PMC traffic files
    are only valid for the code they were taken on (tools/make_traffic.py stores the hash of that
    code). Not part of it: the list of developer switches (env.hpp), the C header, and the
    high-level translation unit (highlevel.hip, huffman.hpp, format.hpp), which launches none of the
    step's kernels."""
    import hashlib
    from mgard_amd import _build
    hsh = hashlib.sha256()
    for f in sorted(_build._closure(os.path.join(ROOT, "mgard_amd", "csrc", "capi.hip"))):
        name = os.path.basename(f)
        if name == "capi.hip" or name == "hierarchy.hpp" or name.startswith("kernels_"):
            hsh.update(name.encode())
            hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "512f32": dict(shape=(512, 512, 512), dtype="float32", s=float("inf"), nonuniform=False,
                   what="3D 512x512x512 float32 uniform grid, REL L-inf tol 1e-3, s=inf"),
    # configs[2]: non-uniform spacing, s = 0 (mass-matrix + tridiagonal path, L2 norm)
    "512f64nu": dict(shape=(512, 512, 512), dtype="float64", s=0.0, nonuniform=True,
                     what="3D 512x512x512 float64 NON-uniform spacing, REL tol 1e-3, s=0"),
    # configs[3]: one rank's 4-D slab of the 64 x 512^3 volume (weak scaling: 8 x 512^3 per GPU)
    "4d": dict(shape=(8, 512, 512, 512), dtype="float32", s=float("inf"), nonuniform=False,
               what="4D 8x512x512x512 float32 slab (one of the 8 slabs of 64x512^3, split on dim 0), "
                    "REL L-inf tol 1e-3, s=inf"),
    # configs[4]: 1024^3 round trip, error against the tolerance, end-to-end GB/s
    "1024f32": dict(shape=(1024, 1024, 1024), dtype="float32", s=float("inf"), nonuniform=False,
                    what="3D 1024x1024x1024 float32 uniform grid, REL L-inf tol 1e-3, s=inf"),
}


def _last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


LAUNCH_TIMEOUT_S = 1800   # a hung rank (collective left by a failed peer) must not hang the driver
COLLECTIVE_TIMEOUT_S = 300


def launch_ranks(args, argv):
    """N > 1 without WORLD_SIZE: this process never initialises a GPU. It starts the ranks with
    torch.distributed.run (fresh child processes), relays rank 0's JSON line, then runs the
    one-process / N-devices leg (mgh_compress_multi) in another child and attaches its result."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = None
    for attempt in range(3):
        # (a port found by bind-then-close can be taken by the time the ranks rendezvous when several
        # launches run side by side: retry on a fresh one if the rendezvous fails to bind)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + argv
        try:
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                               timeout=LAUNCH_TIMEOUT_S)
        except subprocess.TimeoutExpired as e:
            sys.stderr.write("bench.py: the ranks did not finish within %d s\n%s\n" % (
                LAUNCH_TIMEOUT_S, (e.stderr or "")[-2000:] if isinstance(e.stderr, str) else ""))
            raise SystemExit(2)
        if p.returncode != 0 and "Address already in use" in p.stderr and attempt < 2:
            continue
        break
    sys.stderr.write(p.stderr[-4000:])
    res = _last_json_line(p.stdout)
    if p.returncode != 0 or res is None:
        sys.stdout.write(p.stdout[-4000:])
        raise SystemExit(p.returncode or 1)
    if not args.dist_dry_run and not args.no_native_multi:
        cmd2 = [sys.executable, os.path.abspath(__file__), "--native-multi", "--gpus", str(args.gpus)]
        try:
            p2 = subprocess.run(cmd2, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                text=True, timeout=600)
            r2 = _last_json_line(p2.stdout)
            res["native_multi"] = r2 if (p2.returncode == 0 and r2) else {
                "error": "rc %d: %s" % (p2.returncode, p2.stderr[-300:])}
        except subprocess.TimeoutExpired:
            res["native_multi"] = {"error": "timeout"}
        # ... and the C entry points for one rank per GPU (RCCL inside the library), one thread per rank
        cmd3 = [sys.executable, os.path.abspath(__file__), "--capi-dist", "--gpus", str(args.gpus)]
        try:
            p3 = subprocess.run(cmd3, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                text=True, timeout=240)
            r3 = _last_json_line(p3.stdout)
            res["capi_dist"] = r3 if (p3.returncode == 0 and r3) else {
                "error": "rc %d: %s" % (p3.returncode, p3.stderr[-300:])}
        except subprocess.TimeoutExpired:
            res["capi_dist"] = {"error": "timeout"}
    print(json.dumps(res), flush=True)


def dist_dry_run(args):
    """GPU-less check of the N > 1 control flow (tests/test_distributed_cpu.py): the ranks the
    launcher started form a gloo group, exchange the scalar norm exactly like the GPU path (MAX
    all-reduce), count each other, and rank 0 prints a line of the usual shape."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    nrm = torch.tensor([1.0 + rank], dtype=torch.float64)
    seen = torch.ones(1, dtype=torch.int64)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dist.all_reduce(nrm, op=dist.ReduceOp.MAX)
    dist.barrier()
    el = time.perf_counter() - t0
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    ok = float(nrm.item()) == float(world)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (gloo, no GPU): launcher + norm exchange only", "value": 0.0,
                          "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / max(args.steps, 1) * 1e3, 4), "dry_run": True,
                          "rccl_ranks_seen": int(seen.item()), "norm_exchange_ok": ok}), flush=True)
    if not ok:
        raise SystemExit(3)


def native_multi(args):
    """One process, N devices: mgh_compress_multi / mgh_decompress_multi (include/mgard_hip_compress.h)
    on a host volume of N slabs of 128 x 512 x 512 f32 -- host buffers in and out, so the figure is
    PCIe-inclusive and informational; the point of the leg is that the native path runs on N
    physical devices and reconstructs within the bound."""
    import numpy as np
    import torch
    from mgard_amd import highlevel as hl
    import mgard_amd
    from tests.util import smooth_field
    n = min(args.gpus, torch.cuda.device_count())
    u = smooth_field((128 * n, 512, 512), np.float32)
    devs = tuple(range(n))
    buf = hl.compress_multi(u, TOL, float("inf"), mgard_amd.REL, devices=devs)
    t0 = time.perf_counter()
    buf = hl.compress_multi(u, TOL, float("inf"), mgard_amd.REL, devices=devs)
    c_s = time.perf_counter() - t0
    v = hl.decompress_multi(buf, devices=devs)
    t1 = time.perf_counter()
    v = hl.decompress_multi(buf, devices=devs)
    d_s = time.perf_counter() - t1
    err = float(np.max(np.abs(v.astype(np.float64) - u)))
    nrm = float(np.max(np.abs(u)))
    res = {"what": "mgh_compress_multi / mgh_decompress_multi: one process, one host thread per "
                   "device, host buffers (PCIe-inclusive, informational)",
           "devices": list(devs), "shape": list(u.shape),
           "compress_GBps": round(u.nbytes / c_s / 1e9, 3),
           "decompress_GBps": round(u.nbytes / d_s / 1e9, 3),
           "compression_ratio": round(u.nbytes / buf.size, 3),
           "within_tolerance": bool(err <= TOL * nrm)}
    # the same volume device-resident on devices[0]: slabs of the other devices travel by
    # hipMemcpyPeerAsync (xGMI), the container comes back in device memory of devices[0]
    try:
        d_u = torch.from_numpy(u).to("cuda:%d" % devs[0])
        dbuf = hl.compress_multi(d_u, TOL, float("inf"), mgard_amd.REL, devices=devs)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        dbuf = hl.compress_multi(d_u, TOL, float("inf"), mgard_amd.REL, devices=devs)
        torch.cuda.synchronize()
        dc_s = time.perf_counter() - t2
        res["device_resident"] = {"compress_GBps": round(u.nbytes / dc_s / 1e9, 3),
                                  "same_container_as_host_input": bool(np.array_equal(dbuf.cpu().numpy(), buf))}
    except (mgard_amd.MgardHipError, RuntimeError) as e:
        res["device_resident"] = {"error": str(e)[:300]}
    print(json.dumps(res), flush=True)


def capi_dist(args):
    """configs[3] through the C entry points for one rank per GPU (mgh_compress_dist /
    mgh_decompress_dist, include/mgard_hip_compress.h): one host thread per device stands in for a rank
    (its own RCCL communicator rank, its own 8 x 512^3 f32 slab resident on its device); the norm
    all-reduce, the record gather to rank 0 and the hand-out on the way back run over RCCL inside the
    library. Weak scaling: the domain is (8 N) x 512^3. Own child process with a time limit: a failure
    here cannot touch the metric's line."""
    import ctypes as C
    import threading
    import numpy as np
    import torch
    import mgard_amd
    from mgard_amd import highlevel as hl
    n = min(args.gpus, torch.cuda.device_count())
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        path = "librccl.so.1"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid = UniqueId()
    if rccl.ncclGetUniqueId(C.byref(uid)) != 0:
        raise SystemExit("ncclGetUniqueId failed")
    slab = (8, 512, 512, 512)
    slab_bytes = int(np.prod(slab)) * 4
    bar = threading.Barrier(n)
    out, errors, times = {}, [], {}

    def rank_thread(r):
        try:
            torch.cuda.set_device(r)
            dev = torch.device("cuda", r)
            comm = C.c_void_p()
            if rccl.ncclCommInitRank(C.byref(comm), n, uid, r) != 0:
                raise RuntimeError("ncclCommInitRank failed")
            base = gpu_field(torch, slab[1:], torch.float32, dev, seed=20260101 + r)
            d = torch.stack([base * (1.0 + 0.002 * t) + 1e-4 * t for t in range(slab[0])])
            del base
            nrm = float(d.abs().max().item())
            cfg = hl.Config(dev_id=r)
            c = hl.compress_dist(comm.value, r, n, d, TOL, float("inf"), mgard_amd.REL, config=cfg, rccl_path=path)
            torch.cuda.synchronize()
            bar.wait()
            t0 = time.perf_counter()
            c = hl.compress_dist(comm.value, r, n, d, TOL, float("inf"), mgard_amd.REL, config=cfg)
            torch.cuda.synchronize()
            bar.wait()
            t1 = time.perf_counter()
            v = hl.decompress_dist(comm.value, r, n, c, d.shape, d.dtype, config=cfg, device=dev)
            torch.cuda.synchronize()
            bar.wait()
            t2 = time.perf_counter()
            v = hl.decompress_dist(comm.value, r, n, c, d.shape, d.dtype, config=cfg, device=dev)
            torch.cuda.synchronize()
            bar.wait()
            t3 = time.perf_counter()
            times[r] = (t1 - t0, t3 - t2)
            out[r] = (float((v - d).abs().max().item()), nrm, None if c is None else int(c.numel()))
            rccl.ncclCommDestroy(comm)
            hl.release_cache()
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d: %r" % (r, e))
            bar.abort()

    th = [threading.Thread(target=rank_thread, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        print(json.dumps({"error": "; ".join(errors)[:500]}), flush=True)
        return
    gnorm = max(o[1] for o in out.values())
    c_s, d_s = max(t[0] for t in times.values()), max(t[1] for t in times.values())
    print(json.dumps({
        "what": "mgh_compress_dist / mgh_decompress_dist: one host thread per device as a rank, RCCL inside the "
                "library (norm all-reduce, record gather / hand-out); one 8x512^3 f32 slab per rank, device-resident",
        "ranks": n, "domain": [8 * n, 512, 512, 512],
        "compress_ms": round(c_s * 1e3, 3), "compress_GBps": round(n * slab_bytes / c_s / 1e9, 2),
        "decompress_ms": round(d_s * 1e3, 3), "decompress_GBps": round(n * slab_bytes / d_s / 1e9, 2),
        "container_bytes": out[0][2], "compression_ratio": round(n * slab_bytes / out[0][2], 3),
        "roundtrip_linf_error": max(o[0] for o in out.values()), "tolerance_abs": TOL * gnorm,
        "within_tolerance": bool(max(o[0] for o in out.values()) <= TOL * gnorm)}), flush=True)


def gpu_field(torch, shape, dtype, dev, seed=20260101):
    """The recipe of tests/util.smooth_field evaluated on the device (the legs that are not
    compared with the CPU oracle do not need the host copy; 1024^3 takes ~20 s in numpy)."""
    D = len(shape)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ax = []
    for d, n in enumerate(shape):
        x = torch.arange(n, dtype=torch.float64, device=dev) / max(n - 1, 1)
        ax.append(x.reshape([n if k == d else 1 for k in range(D)]))
    f = [3.0, 2.0, 5.0, 1.0, 4.0]
    two_pi = 2 * 3.141592653589793
    u = torch.sin(two_pi * f[0] * ax[D - 1]).to(dtype)
    if D >= 2:
        u = u * torch.cos(two_pi * f[1] * ax[D - 2]).to(dtype)
    if D >= 3:
        u = u + (0.5 * torch.sin(two_pi * f[2] * ax[D - 3])).to(dtype)
    for d in range(D - 3):
        u = u + (0.25 * torch.cos(two_pi * f[3 + (d % 2)] * ax[d])).to(dtype)
    u = u.expand(shape).contiguous()
    noise = torch.rand(shape, generator=g, dtype=dtype, device=dev)
    u.add_(noise.mul_(2e-3).sub_(1e-3))
    return u


def config_leg(torch, mgard_amd, name, dev, local_rank, steps=5, end_to_end=False, dist=None, world=1,
               rank=0, host_e2e=False):
    """One of the other BASELINE.json configurations as a step-only leg of the default run
    (`other_configs`): input generated on the device, `steps` timed steps after 2 warm-ups."""
    import numpy as np
    from tests.util import nonuniform_coords
    cfg = CONFIGS[name]
    shape = cfg["shape"]
    np_dt = np.dtype(cfg["dtype"])
    t_dt = torch.float32 if np_dt.itemsize == 4 else torch.float64
    S = cfg["s"]
    coords = nonuniform_coords(shape, np_dt) if cfg["nonuniform"] else None
    setup_error = None
    try:
        if len(shape) == 4:
            base = gpu_field(torch, shape[1:], t_dt, dev, seed=20260101 + rank)
            d_u = torch.stack([base * (1.0 + 0.002 * t) + 1e-4 * t for t in range(shape[0])])
            del base
        else:
            d_u = gpu_field(torch, shape, t_dt, dev)
        h = mgard_amd.Hierarchy(shape, np_dt, coords=coords, device=local_rank)
        N = h.total
        cap = N // (16 if len(shape) == 3 else 8)
        q = torch.empty(shape, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        oidx = torch.empty(cap, dtype=torch.int64, device=dev)
        oval = torch.empty(cap, dtype=torch.int64, device=dev)
        bufs = (q, cnt, oidx, oval)
        nrm_t = torch.zeros(1, dtype=h.torch_dtype, device=dev)
    except (mgard_amd.MgardHipError, RuntimeError) as e:
        if dist is None:
            raise
        setup_error = str(e)[:200]
    if dist is not None:
        # a rank that could not set up must not leave the others inside the collectives of the step
        okt = torch.tensor([0 if setup_error else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if not bool(okt.item()):
            return {"error": "set-up failed on some rank: %s" % setup_error}

    def step():
        if dist is None:
            h.decompose_quantize(d_u, mgard_amd.REL, TOL, S, 0.0, bufs=bufs, want_norm=False)
        else:  # one subdomain per rank: global norm by one scalar all-reduce, then an ABS bound
            h.norm_device(d_u, S, out=nrm_t)
            dist.all_reduce(nrm_t, op=dist.ReduceOp.MAX)
            h.decompose_quantize_dn(d_u, mgard_amd.REL, TOL, S, nrm_t, world, bufs)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
    if dist is not None:
        # First step with every rank-local call guarded: all ranks agree (MIN all-reduce) that a phase
        # went through before anyone enters the collective behind it, so a rank that cannot run the
        # step (out of memory, an unsupported shape) does not leave its peers inside an all-reduce
        # until the RCCL time-out. The timed steps below are the plain ones.
        def agree(err):
            okt = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            return bool(okt.item())
        for phase in ("norm", "decompose_quantize"):
            err = None
            try:
                if phase == "norm":
                    h.norm_device(d_u, S, out=nrm_t)
                else:
                    h.decompose_quantize_dn(d_u, mgard_amd.REL, TOL, S, nrm_t, world, bufs)
                torch.cuda.synchronize()
            except (mgard_amd.MgardHipError, RuntimeError) as e:
                err = str(e)[:200]
            if not agree(err):
                h.close()
                return {"error": "%s failed on some rank: %s" % (phase, err)}
            if phase == "norm":
                dist.all_reduce(nrm_t, op=dist.ReduceOp.MAX)
    for _ in range(2):
        step()
    barrier()
    # per-kernel breakdown (HIP events on every launch, untimed), then the timed steps with events
    # on the dominant kernel's launches only -- like the metric's leg
    NPROF = 2
    h.profile(True)
    for _ in range(NPROF):
        step()
    torch.cuda.synchronize()
    prof = h.profile_read(reset=True)
    dominant = max(prof.items(), key=lambda kv: kv[1][0])[0]
    h.profile(True, only=dominant)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    dom_ms, dom_launches = h.profile_read(reset=True)[dominant]
    h.profile(False)
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms = el / steps * 1e3
    in_bytes = N * np_dt.itemsize
    out = {"workload": cfg["what"] + ("" if dist is None else "; one slab per rank, %d ranks (weak scaling)" % world),
           "steps": steps, "ms_per_step": round(ms, 4),
           "value": round(world * in_bytes / ms / 1e6, 2), "unit": "GB/s (input, all ranks)",
           "hbm_frac_whole_step": round((np_dt.itemsize + 8.0) * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "outliers_per_step": int(cnt.item()),
           "data": "synthetic, generated on the device (same recipe as the metric's field)"}
    alg = algorithmic_bytes_per_step(h, np_dt.itemsize)
    achieved = alg.get(dominant, 0) * steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic, traffic_note = pmc_traffic(name, dominant)
    out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                       "traffic_source": traffic_note,
                       "avg_launch_ms": round(dom_ms / max(dom_launches, 1), 5), "launches": dom_launches,
                       "algorithmic_bytes_per_step": alg.get(dominant, 0),
                       "kernel_ms_per_step_all": {k: round(v[0] / NPROF, 4) for k, v in sorted(prof.items())}}
    assert out["outliers_per_step"] <= cap
    if end_to_end:
        # configs[4]: compress + decompress round trip through the container, error against the
        # tolerance, end-to-end GB/s (device-resident in and out)
        from mgard_amd import highlevel
        nrm_host = float(h.norm(d_u, S))
        del q, oidx, oval, bufs
        obuf = torch.empty(in_bytes + 1000000, dtype=torch.uint8, device=dev)
        stream = highlevel.compress(d_u, TOL, S, mgard_amd.REL, out=obuf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        NE = 5
        for _ in range(NE):
            stream = highlevel.compress(d_u, TOL, S, mgard_amd.REL, out=obuf)
        torch.cuda.synchronize()
        c_ms = (time.perf_counter() - t1) / NE * 1e3
        back = torch.empty_like(d_u)
        highlevel.decompress(stream, out=back)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(NE):
            highlevel.decompress(stream, out=back)
        torch.cuda.synchronize()
        x_ms = (time.perf_counter() - t2) / NE * 1e3
        err = float((back - d_u).abs().max().item())
        out["end_to_end"] = {"what": "mgh_compress + mgh_decompress round trip, device-resident, MGARD-X container",
                             "compress_ms": round(c_ms, 3), "compress_GBps": round(in_bytes / c_ms / 1e6, 2),
                             "decompress_ms": round(x_ms, 3), "decompress_GBps": round(in_bytes / x_ms / 1e6, 2),
                             "roundtrip_GBps": round(in_bytes / (c_ms + x_ms) / 1e6, 2),
                             "compression_ratio": round(in_bytes / int(stream.numel()), 3),
                             "roundtrip_linf_error": err, "tolerance_abs": TOL * nrm_host,
                             "within_tolerance": bool(err <= TOL * nrm_host)}
        del back, obuf, stream
        highlevel.release_cache()
        if host_e2e:
            try:
                u_host = d_u.cpu().numpy()
                del d_u
                torch.cuda.empty_cache()
                out["end_to_end_host"] = host_leg(torch, mgard_amd, u_host, TOL, S, nrm_host, reps=2, batches=2)
                del u_host
            except (mgard_amd.MgardHipError, RuntimeError, ValueError) as e:
                out["end_to_end_host"] = {"error": str(e)[:300]}
    h.close()
    return out


def inside_field(torch, shape, dtype, dev, seed=20260102):
    """A field whose quantized coefficients stay inside an 8192-entry dictionary at 1e-3 whatever the
    dimensionality (the quantizer's bins shrink with 1 + 3^D: `gpu_field`'s recipe leaves the
    dictionary in 4-D / 5-D and the step then measures outlier lists): one period of a sine across
    every extent of 32 and more points, 1 % of that across the short ones, 1e-3 uniform noise."""
    import numpy as np
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    u = (torch.rand(shape, dtype=dtype, device=dev, generator=g) * 2 - 1) * 1e-3
    for k, n in enumerate(shape):
        x = torch.arange(n, dtype=torch.float64, device=dev) / max(n - 1, 1)
        view = [1] * len(shape)
        view[k] = n
        u += ((1.0 if n >= 32 else 0.01) * torch.sin(2 * np.pi * x + 0.3 * k)).to(dtype).view(view)
    return u


def beyond_leg(torch, mgard_amd, dev, local_rank, steps=5):
    """Informational, never `value`: the step (norm + decompose + quantize, REL 1e-3, s = inf) and the
    way back on two shapes outside BASELINE.json's list that the reference is used with -- D = 5 (the
    generic N-D kernels) and the shape of its own flagship data set, XGC's 8 x 16395 x 39 x 39 float64
    (examples/mgard-x/CompressXgcData) -- on `inside_field`."""
    import numpy as np
    out = {}
    for name, shape, dt in (("5d_8x8x64x64x64_f32", (8, 8, 64, 64, 64), torch.float32),
                            ("xgc_8x16395x39x39_f64", (8, 16395, 39, 39), torch.float64)):
        np_dt = np.float32 if dt == torch.float32 else np.float64
        u = inside_field(torch, shape, dt, dev)
        h = mgard_amd.Hierarchy(shape, np_dt, device=local_rank)
        N = u.numel()
        q = torch.empty(shape, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        oi = torch.empty(N // 8, dtype=torch.int64, device=dev)
        ov = torch.empty(N // 8, dtype=torch.int64, device=dev)
        # the metric's step: [norm +] decompose + quantize, int64 out into the caller's buffers, no host sync
        fwd = lambda: h.decompose_quantize(u, mgard_amd.REL, 1e-3, float("inf"), 0.0, outlier_cap=N // 8,
                                           bufs=(q, cnt, oi, ov), want_norm=False)
        for _ in range(2):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fwd()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        nb = N * u.element_size()
        n_out = int(cnt.item())
        assert n_out <= N // 8, "outlier buffer too small"
        nrm = h.norm(u, float("inf"))
        back = torch.empty_like(u)
        qq = q.clone()
        bwd = lambda: h.dequantize_recompose(qq, mgard_amd.REL, 1e-3, float("inf"), nrm, outlier_idx=oi[:n_out],
                                             outlier_val=ov[:n_out], out=back)
        for _ in range(2):
            bwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bwd()
        torch.cuda.synchronize()
        msb = (time.perf_counter() - t0) / steps * 1e3
        err = float((back - u).abs().max())
        out[name] = {"shape": list(shape), "what": "[norm +] decompose + quantize, int64 out; back: dequantize + "
                                                     "recompose, int64 in; REL 1e-3, s = inf",
                     "ms_per_step": round(ms, 4), "GBps": round(nb / ms / 1e6, 1),
                     "back_ms": round(msb, 4), "back_GBps": round(nb / msb / 1e6, 1),
                     "outliers": n_out, "l_target": h.l_target,
                     "roundtrip_linf_error": err, "tolerance_abs": 1e-3 * nrm,
                     "within_tolerance": bool(err <= 1e-3 * nrm)}
        h.close()
        del u, back, q, qq, oi, ov
        torch.cuda.empty_cache()
    return out


def host_leg(torch, mgard_amd, u, tol, s, nrm_host, reps=4, batches=3):
    """SURVEY.md section 8(d), the reference's own usage: mgh_compress / mgh_decompress with HOST
    buffers in and out (H2D + every device stage + lossless + D2H inside the timed call). Three kinds
    of caller memory -- pageable, registered by the caller (mgard_x::pin_memory) and
    auto_pin_host_buffers = 1 (the reference's default: the library registers and unregisters inside
    the call) -- next to the box's own pinned H2D / D2H rate for the same bytes, measured here.
    Never `value`: the metric is defined on device-resident input."""
    import ctypes as C
    import numpy as np
    from mgard_amd import highlevel
    in_bytes = u.nbytes
    out = {"what": "mgh_compress / mgh_decompress, host buffers in and out (pre-allocated, touched), "
                   "MGARD-X container (Huffman); best of %d batches of %d calls" % (batches, reps),
           "shape": list(u.shape), "dtype": str(u.dtype), "input_bytes": in_bytes}

    def best(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(batches):
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / reps * 1e3)
        return min(ts), r

    # the link itself: one pinned copy of the same bytes each way (and of the container's bytes)
    pin_t = torch.empty(in_bytes, dtype=torch.uint8, pin_memory=True)
    dev_t = torch.empty(in_bytes, dtype=torch.uint8, device="cuda")
    h2d_ms, _ = best(lambda: dev_t.copy_(pin_t, non_blocking=True))
    d2h_ms, _ = best(lambda: pin_t.copy_(dev_t, non_blocking=True))
    del pin_t, dev_t
    out["link"] = {"pinned_h2d_GBps": round(in_bytes / h2d_ms / 1e6, 2), "pinned_d2h_GBps": round(in_bytes / d2h_ms / 1e6, 2),
                   "pinned_h2d_ms": round(h2d_ms, 3), "pinned_d2h_ms": round(d2h_ms, 3)}
    h2d, d2h = in_bytes / h2d_ms / 1e6, in_bytes / d2h_ms / 1e6

    def rows(kind, cfg, x, cbuf, back):
        c_ms, stream = best(lambda: highlevel.compress(x, tol, s, mgard_amd.REL, config=cfg, out=cbuf))
        x_ms, v = best(lambda: highlevel.decompress(stream, config=cfg, out=back))
        err = float(np.max(np.abs(v - u))) if s == float("inf") else float(np.sqrt(np.mean((v.astype(np.float64) - u) ** 2)))
        cb = int(stream.size)
        # what the call cannot go below with ONE subdomain under a REL bound: all of the input on the
        # link, then (the norm needs every byte) the record back -- and the mirror image for decompression
        floor_ms = in_bytes / h2d / 1e6 + cb / d2h / 1e6
        floor_x_ms = cb / h2d / 1e6 + in_bytes / d2h / 1e6
        out[kind] = {"compress_ms": round(c_ms, 3), "compress_GBps": round(in_bytes / c_ms / 1e6, 2),
                     "compress_frac_of_h2d_rate": round(in_bytes / c_ms / 1e6 / h2d, 3),
                     "compress_frac_of_link_floor": round(floor_ms / c_ms, 3),
                     "decompress_ms": round(x_ms, 3), "decompress_GBps": round(in_bytes / x_ms / 1e6, 2),
                     "decompress_frac_of_d2h_rate": round(in_bytes / x_ms / 1e6 / d2h, 3),
                     "decompress_frac_of_link_floor": round(floor_x_ms / x_ms, 3),
                     "container_bytes": cb, "within_tolerance": bool(err <= tol * nrm_host)}
        return stream

    cbuf = np.zeros(in_bytes + 1000000, dtype=np.uint8)   # (touched: no first-touch faults inside the calls)
    back = np.zeros(u.shape, dtype=u.dtype)
    stream = rows("pageable", highlevel.Config(), u, cbuf, back)
    ref_stream = bytes(stream[:4096])
    # auto_pin_host_buffers = 1 on the same pageable arrays
    try:
        rows("auto_pin", highlevel.Config(auto_pin_host_buffers=1), u, cbuf, back)
    except mgard_amd.MgardHipError as e:
        out["auto_pin"] = {"error": str(e)[:200]}
    # registered by the caller: input, container and output
    try:
        t0 = time.perf_counter()
        for a in (u, cbuf, back):
            highlevel.pin(a)
        out["caller_pinned_register_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        try:
            stream = rows("caller_pinned", highlevel.Config(), u, cbuf, back)
            out["caller_pinned"]["same_container_head_as_pageable"] = bool(bytes(stream[:4096]) == ref_stream)
        finally:
            for a in (u, cbuf, back):
                highlevel.unpin(a)
    except mgard_amd.MgardHipError as e:
        out["caller_pinned"] = {"error": str(e)[:200]}
    # output allocated BY THE LIBRARY (output_pre_allocated = 0, the reference's plain call): fresh pages
    try:
        L = highlevel._hl()
        cfg = highlevel.Config()
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        sbuf = np.ascontiguousarray(stream)

        ts, free_ms = [], []
        for _ in range(batches + 1):   # (first batch = warm-up)
            ptrs = []
            t0 = time.perf_counter()
            for _ in range(reps):
                optr = C.c_void_p()
                mgard_amd._check(L.mgh_decompress(C.c_void_p(sbuf.ctypes.data), sbuf.size, C.byref(optr), C.byref(cfg), 0))
                ptrs.append(optr)
            ts.append((time.perf_counter() - t0) / reps * 1e3)
            t0 = time.perf_counter()
            for optr in ptrs:   # (the caller's free() is not part of the call: timed beside it)
                libc.free(optr)
            free_ms.append((time.perf_counter() - t0) / reps * 1e3)
        x_ms = min(ts[1:])
        out["library_allocated_output"] = {"decompress_ms": round(x_ms, 3), "decompress_GBps": round(in_bytes / x_ms / 1e6, 2),
                                           "callers_free_ms": round(min(free_ms[1:]), 3)}
    except (mgard_amd.MgardHipError, OSError) as e:
        out["library_allocated_output"] = {"error": str(e)[:200]}
    highlevel.release_cache()
    return out


def volume_leg(torch, mgard_amd, dev):
    """BASELINE.json configs[3] as ONE volume on one GPU: 64 x 512^3 f32 (34 GB, device-resident),
    `Variable` decomposition {8} x 8 on dim 0 (the 8 slabs the 8 ranks of the weak-scaling run own)
    through mgh_compress / mgh_decompress: 8 records behind one header, REL bound against the norm
    of the whole volume, round trip checked. End to end (Huffman stage and container included)."""
    from mgard_amd import highlevel
    free, _ = torch.cuda.mem_get_info()
    if free < 140e9:
        return {"skipped": "needs ~130 GB of free device memory, %.0f GB free" % (free / 1e9)}
    base = gpu_field(torch, (512, 512, 512), torch.float32, dev)
    vol = torch.empty((64, 512, 512, 512), dtype=torch.float32, device=dev)
    for t in range(64):
        vol[t] = base * (1.0 + 0.002 * t) + 1e-4 * t
    del base
    in_bytes = vol.numel() * 4
    cfg = highlevel.Config(domain_decomposition=highlevel.DD_VARIABLE, domain_decomposition_dim=0,
                           domain_decomposition_sizes=[8] * 8)
    obuf = torch.empty(in_bytes // 2, dtype=torch.uint8, device=dev)
    stream = highlevel.compress(vol, TOL, float("inf"), mgard_amd.REL, config=cfg, out=obuf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stream = highlevel.compress(vol, TOL, float("inf"), mgard_amd.REL, config=cfg, out=obuf)
    torch.cuda.synchronize()
    c_ms = (time.perf_counter() - t0) * 1e3
    back = torch.empty_like(vol)
    highlevel.decompress(stream, out=back, config=cfg)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    highlevel.decompress(stream, out=back, config=cfg)
    torch.cuda.synchronize()
    x_ms = (time.perf_counter() - t1) * 1e3
    nrm = float(vol.abs().max().item())
    err = max(float((back[t] - vol[t]).abs().max().item()) for t in range(64))
    out = {"workload": "4D 64x512x512x512 float32 (34 GB) as ONE device-resident volume, Variable decomposition "
                       "{8} x 8 on dim 0, mgh_compress / mgh_decompress (Huffman, container), REL 1e-3 "
                       "against the norm of the whole volume",
           "compress_ms": round(c_ms, 2), "compress_GBps": round(in_bytes / c_ms / 1e6, 2),
           "decompress_ms": round(x_ms, 2), "decompress_GBps": round(in_bytes / x_ms / 1e6, 2),
           "compression_ratio": round(in_bytes / int(stream.numel()), 3), "subdomains": 8,
           "roundtrip_linf_error": err, "tolerance_abs": TOL * nrm, "within_tolerance": bool(err <= TOL * nrm)}
    del back, vol, obuf, stream
    highlevel.release_cache()
    torch.cuda.empty_cache()
    return out


def scatter_gather_leg(torch, mgard_amd, dist, dev, local_rank, rank, world, slab_t=8):
    """N > 1: BASELINE.json configs[3] the way north_star words it -- one rank holds the whole
    (slab_t * N) x 512^3 volume device-resident; block scatter of the slabs of the slowest dimension
    over RCCL point-to-point sends (mgard_amd.distributed.scatter_slabs), global norm by one scalar
    MAX all-reduce, every rank compresses its slab through mgh_compress with the local ABS bound
    (calc_local_abs_tol), payload gather to rank 0 (gather_payloads), ONE container behind a header
    that declares the decomposition; rank 0 opens it with mgh_decompress and checks the bound.
    Every phase timed on its own (barrier + synchronize around it, MAX over ranks)."""
    import numpy as np
    from mgard_amd import distributed as mdist
    from mgard_amd import highlevel
    shape = (slab_t * world, 512, 512, 512)
    slab_shape = (slab_t,) + shape[1:]
    slab_bytes = int(np.prod(slab_shape)) * 4

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def agree(ok):
        # every rank learns whether ALL ranks got through the phase: a rank that failed must not leave
        # the others inside a collective
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    vol, err = None, None
    try:
        if rank == 0:
            base = gpu_field(torch, shape[1:], torch.float32, dev)
            vol = torch.empty(shape, dtype=torch.float32, device=dev)
            for t in range(shape[0]):
                vol[t] = base * (1.0 + 0.002 * t) + 1e-4 * t
            del base
        obuf = torch.empty(slab_bytes // 2 + 1000000, dtype=torch.uint8, device=dev)
    except RuntimeError as e:
        err = str(e)[:200]
    if not agree(err is None):
        return {"error": "set-up failed on some rank: %s" % err}
    cfg = highlevel.Config(dev_id=local_rank)
    times = {}

    def timed(name, fn):
        sync()
        t0 = time.perf_counter()
        out = fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times[name] = float(t.item())
        return out

    class PhaseError(Exception):
        pass

    def local(fn):
        """Rank-local work of a phase. Every rank learns whether ALL ranks got through it BEFORE anyone
        enters the next collective: a rank that failed must not leave its peers inside an all-reduce or
        a receive until the RCCL time-out."""
        res, err = None, None
        try:
            res = fn()
        except (mgard_amd.MgardHipError, RuntimeError, AssertionError, ValueError) as e:
            err = "%s: %s" % (type(e).__name__, str(e)[:200])
        if not agree(err is None):
            raise PhaseError(err or "a peer rank failed")
        return res

    def round_trip():
        slab = timed("scatter", lambda: mdist.scatter_slabs(vol, shape, src=0, device=dev, dtype=torch.float32))
        nrm_t = torch.zeros(1, dtype=torch.float32, device=dev)

        def norm():
            def mine():
                h = mgard_amd.Hierarchy(slab_shape, np.float32, device=local_rank)
                h.norm_device(slab, float("inf"), out=nrm_t)
                h.close()
            local(mine)
            dist.all_reduce(nrm_t, op=dist.ReduceOp.MAX)
            return float(nrm_t.item())
        nrm = timed("norm_exchange", norm)
        # (in the arithmetic of the data type, like calc_local_abs_tol: the reader recomputes it from
        # the header as float(tol) * float(norm))
        atol = float(np.float32(TOL) * np.float32(nrm))

        def compress():
            assert abs(atol - mdist.local_abs_tol(mdist.REL, nrm, TOL, float("inf"), world)) <= 1e-6 * atol
            stream = highlevel.compress(slab, atol, float("inf"), mgard_amd.ABS, config=cfg, out=obuf)
            ms = highlevel.metadata_parse(bytes(stream[:8192].cpu().numpy()))["metadata_size"]
            return stream[ms + 8:]        # the subdomain's record without its own size prefix
        record = timed("compress", lambda: local(compress))
        payloads = timed("gather", lambda: mdist.gather_payloads(record, dst=0))
        return slab, nrm, payloads

    try:
        slab, nrm, payloads = round_trip()     # warm-up: allocations, hierarchies, RCCL channels
        del slab, payloads
        slab, nrm, payloads = round_trip()
    except PhaseError as e:
        del vol, obuf
        highlevel.release_cache()
        torch.cuda.empty_cache()
        return {"error": "a phase failed on some rank: %s" % e}
    out = {"workload": "4D %dx512x512x512 float32 device-resident on rank 0: RCCL block scatter -> norm all-reduce "
                       "-> mgh_compress per slab -> RCCL payload gather -> one container" % shape[0],
           "ranks": world, "volume_GB": round(world * slab_bytes / 1e9, 2)}
    for k, v in times.items():
        out[k + "_ms"] = round(v * 1e3, 3)
    moved = (world - 1) * slab_bytes
    out["scatter_GBps"] = round(moved / max(times["scatter"], 1e-9) / 1e9, 2)
    total = sum(times.values())
    out["end_to_end_ms"] = round(total * 1e3, 3)
    out["end_to_end_GBps"] = round(world * slab_bytes / total / 1e9, 2)
    out["compress_only_GBps"] = round(world * slab_bytes / max(times["compress"], 1e-9) / 1e9, 2)
    ok, detail = True, None
    if rank == 0:
        try:
            header = highlevel.metadata_serialize(mgard_amd.FLOAT, list(shape),
                                                  mgard_amd.REL, TOL, float("inf"), norm=nrm,
                                                  dd=(highlevel.DD_MAXDIM, 0, slab_t))
            container = mdist.assemble_container(header, payloads)
            out["container_bytes"] = int(container.numel())
            out["compression_ratio"] = round(world * slab_bytes / int(container.numel()), 3)
            back = torch.empty(shape, dtype=torch.float32, device=dev)
            highlevel.decompress(container, out=back, config=cfg)
            e = max(float((back[t] - vol[t]).abs().max().item()) for t in range(shape[0]))
            out["roundtrip_linf_error"], out["tolerance_abs"] = e, TOL * nrm
            out["within_tolerance"] = bool(e <= TOL * nrm)
            del back, container
        except (mgard_amd.MgardHipError, RuntimeError) as e:
            ok, detail = False, str(e)[:300]
    if not agree(ok):
        out["error"] = detail or "rank 0 could not open the container"
    del vol, slab, payloads, obuf
    highlevel.release_cache()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: a timed window of ~90 ms -- a fresh box was seen to run its first 25 ms at 2/3 speed once)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-step", action="store_true",
                    help="time the step only: no decompression / 16-bit / end-to-end legs, no CPU "
                         "baseline (profiling runs: every launch of the trace belongs to the metric)")
    ap.add_argument("--shape", type=str, default=None, help="override, e.g. 256,256,256")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="512f32",
                    help="BASELINE.json configuration: 512f32 = configs[1] (the metric's, default), "
                         "512f64nu = configs[2], 1024f32 = configs[4]")
    ap.add_argument("--no-host-leg", action="store_true",
                    help="skip the host-buffer end-to-end leg (end_to_end_host)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the step-only legs of configs[2], [3], [4] in the default run")
    ap.add_argument("--no-native-multi", action="store_true",
                    help="N > 1 launcher: skip the one-process / N-devices leg (mgh_compress_multi)")
    ap.add_argument("--native-multi", action="store_true",
                    help="run ONLY the one-process / N-devices leg (what the launcher starts as its second child)")
    ap.add_argument("--capi-dist", action="store_true",
                    help="(internal) run only the mgh_compress_dist leg: one thread per device as a rank")
    ap.add_argument("--dist-dry-run", action="store_true",
                    help="GPU-less check of the N > 1 launcher: gloo ranks, norm exchange only")
    ap.add_argument("--sg-slab", type=int, default=8,
                    help="N > 1 scatter_gather leg: time steps (dim 0) per rank, default 8 (= 8 x 512^3 per rank)")
    ap.add_argument("--force-dist-path", action="store_true",
                    help="exercise the N>1 code path (process group + norm all-reduce) with any "
                         "world size, e.g. 1 (developer check)")
    args = ap.parse_args()

    # ---- N > 1 as the driver calls it (`python3 bench.py --gpus N ...`): become the launcher
    # BEFORE anything imports torch or touches a GPU ----
    if args.native_multi:
        return native_multi(args)
    if args.capi_dist:
        return capi_dist(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    if args.dist_dry_run:
        return dist_dry_run(args)

    import numpy as np
    import torch
    import mgard_amd
    from mgard_amd import distributed as mdist
    from tests.util import smooth_field, nonuniform_coords

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.force_dist_path:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cfg = CONFIGS[args.config]
    shape = tuple(int(x) for x in args.shape.split(",")) if args.shape else cfg["shape"]
    np_dt = np.dtype(cfg["dtype"])
    esz = np_dt.itemsize
    S = cfg["s"]
    coords = nonuniform_coords(shape, np_dt) if cfg["nonuniform"] else None

    # synthetic subdomain of this rank (seeded)
    if len(shape) == 4:
        # a slab of time steps: a 3-D field that drifts slowly along dim 0
        base = smooth_field(shape[1:], np_dt.type, seed=20260101 + rank)
        u = np.stack([base * np_dt.type(1.0 + 0.002 * t) + np_dt.type(1e-4 * t) for t in range(shape[0])])
        del base
    else:
        u = smooth_field(shape, np_dt.type, seed=20260101 + rank)
    d_u = torch.from_numpy(u).to(dev)
    h = mgard_amd.Hierarchy(shape, np_dt, coords=coords, device=local_rank)
    N = h.total
    cap = N // (16 if len(shape) == 3 else 8)  # outlier capacity; checked below
    q = torch.empty(shape, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    oidx = torch.empty(cap, dtype=torch.int64, device=dev)
    oval = torch.empty(cap, dtype=torch.int64, device=dev)
    bufs = (q, cnt, oidx, oval)
    nrm_t = torch.zeros(1, dtype=h.torch_dtype, device=dev)

    def step():
        if dist is None:
            # REL bound: norm computed inside the call, on the device (no host round trip)
            return h.decompose_quantize(d_u, mgard_amd.REL, TOL, S, 0.0, bufs=bufs,
                                        want_norm=False)[4]
        # decomposed domain: global norm = MAX of subdomain norms (one scalar all-reduce over
        # RCCL), then an ABS bound per subdomain (mgard_amd/distributed.py)
        h.norm_device(d_u, S, out=nrm_t)
        dist.all_reduce(nrm_t, op=dist.ReduceOp.MAX)
        h.decompose_quantize_dn(d_u, mgard_amd.REL, TOL, S, nrm_t, world, bufs)
        return None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed), then 3 untimed steady-state steps with every kernel bracketed by
    # HIP events: per-kernel breakdown + which kernel dominates ----
    for _ in range(max(args.warmup, 1)):
        nrm = step()
    torch.cuda.synchronize()
    n_out = int(cnt.item())
    assert n_out <= cap, "outlier buffer too small: %d > %d" % (n_out, cap)
    NPROF = 3
    h.profile(True)
    for _ in range(NPROF):
        step()
    torch.cuda.synchronize()
    prof = h.profile_read(reset=True)
    dominant = max(prof.items(), key=lambda kv: kv[1][0])[0]

    # ---- timed region: K steps, HIP events on the dominant kernel's launches only ----
    h.profile(True, only=dominant)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    dom = h.profile_read(reset=True)[dominant]
    h.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HBM traffic of the dominant kernel from the committed PMC passes (same workload only)
    # (a file taken on other kernel sources than the ones this run loads is refused)
    traffic, traffic_note = (None, "no PMC traffic file for this configuration") if args.shape else \
        pmc_traffic(args.config, dominant)
    in_bytes = N * esz
    value = in_bytes * args.steps * world / elapsed / 1e9
    alg = algorithmic_bytes_per_step(h, esz)
    dom_ms, dom_launches = dom
    dom_bytes = alg.get(dominant, 0) * args.steps
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    result = {
        "metric": "compress GB/s (input) at rel-Linf 1e-3, 3D 512^3 f32, decompose+quantize on "
                  "device-resident data" if args.config == "512f32" else
                  "compress GB/s (input), %s, decompose+quantize on device-resident data" % args.config,
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if esz == 4 else "f64",
        "data": "synthetic (seeded smooth field + 1e-3 uniform noise), HBM-resident",
        "config": {"workload": "%s; [norm+]decompose+quantize (int64 out)" % (
                       cfg["what"] if not args.shape else "3D %s %s" % ("x".join(map(str, shape)), np_dt.name)),
                   "name": args.config,
                   "per_gpu_shape": list(shape), "l_target": h.l_target, "dict_size": 8192,
                   "outliers_per_step": n_out,
                   "parallelism": "1 subdomain per GPU%s" % (
                       "" if dist is None else ", scalar norm all-reduce over RCCL")},
        "hbm_frac_whole_step": round((esz + 8.0) * N / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": {"bound": "hbm", "kernel": dominant,
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_note,
                     "avg_launch_ms": round(dom_ms / max(dom_launches, 1), 5),
                     "launches": dom_launches,
                     "algorithmic_bytes_per_step": alg.get(dominant, 0),
                     "kernel_ms_per_step_all": {k: round(v[0] / NPROF, 4)
                                                for k, v in sorted(prof.items())}},
    }
    if dist is None and not args.only_step and len(shape) == 3:
        # How fast does THIS device move the dominant pass's bytes when nothing else is going on?
        # A pure stream with the same read/write mix (N elements read, N int64 written with
        # streaming stores, N/4 elements of side output), no halo, no arithmetic. `frac` above stays
        # priced against the 8 TB/s pin rate; this says how much of the distance to it is the
        # memory system's and how much the kernel's (tools/micro/rw_mix.hip has more variants).
        import ctypes as C
        L = mgard_amd.load_library()
        side = torch.empty(N // 4 + 8, dtype=d_u.dtype, device=dev)
        qcal = torch.empty(N, dtype=torch.int64, device=dev)
        ms = C.c_double(0)
        rc = L.mgh_stream_calibrate(0 if esz == 4 else 1, d_u.data_ptr(), qcal.data_ptr(), side.data_ptr(), N,
                                    10, C.byref(ms), torch.cuda.current_stream().cuda_stream)
        if rc == 0 and ms.value > 0:
            sbytes = N * esz + N * 8 + (N // 4) * esz
            srate = sbytes / (ms.value * 1e-3) / 1e9
            result["roofline"]["stream_calibration"] = {
                "what": "pure stream with the pass's read/write mix (mgh_stream_calibrate): %d B read + %d B "
                        "int64 streaming stores + 2 side arrays per element, no halo, no arithmetic" % (esz, 8),
                "ms": round(ms.value, 4), "GB/s": round(srate, 1),
                "frac_of_peak": round(srate / HBM_PEAK_GBS, 4),
                "kernel_vs_stream_time": round((dom_ms / max(dom_launches, 1)) / ms.value, 3)}
        del side, qcal
    if dist is None and not args.only_step and args.config == "512f32":
        # Informational, never `value`: two volumes in flight on two streams (two hierarchies, two
        # sets of buffers) -- what a pipeline over many subdomains / time steps does. The chain
        # below the top level is latency-bound and leaves the chip idle; the other stream's norm
        # and top-level passes fill it.
        try:
            h2 = mgard_amd.Hierarchy(shape, np_dt, coords=coords, device=local_rank)
            bufs2 = (torch.empty(shape, dtype=torch.int64, device=dev),
                     torch.zeros(1, dtype=torch.int64, device=dev),
                     torch.empty(cap, dtype=torch.int64, device=dev),
                     torch.empty(cap, dtype=torch.int64, device=dev))
            d_u2 = d_u.clone()
            sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

            def two(n):
                for i in range(n):
                    with torch.cuda.stream(sa if i % 2 == 0 else sb):
                        if i % 2 == 0:
                            h.decompose_quantize(d_u, mgard_amd.REL, TOL, S, 0.0, bufs=bufs, want_norm=False)
                        else:
                            h2.decompose_quantize(d_u2, mgard_amd.REL, TOL, S, 0.0, bufs=bufs2, want_norm=False)
            two(4)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            NP2 = 2 * max(args.steps // 2, 2)
            two(NP2)
            torch.cuda.synchronize()
            p_ms = (time.perf_counter() - tp) / NP2 * 1e3
            same2 = bool(torch.equal(bufs2[0], q))
            result["two_streams"] = {"what": "two volumes in flight on two streams (informational: a pipeline "
                                             "over many volumes; `value` is the single-stream step)",
                                     "ms_per_step": round(p_ms, 4), "value": round(in_bytes / p_ms / 1e6, 3),
                                     "unit": "GB/s (input)", "equals_single_stream_output": same2}
            del bufs2, d_u2
            h2.close()
        except mgard_amd.MgardHipError as e:
            result["two_streams"] = {"error": str(e)}
    if args.only_step:
        args.no_cpu_baseline = True
    if dist is None and not args.only_step:
        # decompression side (BASELINE.json configs[4] asks for the round trip): dequantize +
        # recompose of the same volume, error against the requested tolerance
        n_keep = int(cnt.item())
        nrm_host = float(h.norm(d_u, S))
        back = torch.empty_like(d_u)
        q_work = q.clone()
        for _ in range(2):
            h.dequantize_recompose(q_work, mgard_amd.REL, TOL, S, nrm_host,
                                   outlier_idx=oidx[:n_keep], outlier_val=oval[:n_keep], out=back)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ND = 10
        for _ in range(ND):
            h.dequantize_recompose(q_work, mgard_amd.REL, TOL, S, nrm_host,
                                   outlier_idx=oidx[:n_keep], outlier_val=oval[:n_keep], out=back)
        torch.cuda.synchronize()
        d_ms = (time.perf_counter() - t1) / ND * 1e3
        if S == float("inf"):
            err = float((back - d_u).abs().max().item())
            ename = "roundtrip_linf_error"
        else:  # s = 0: the bound is on the L2 norm (root mean square with normalised coordinates)
            err = float(((back - d_u).double() ** 2).mean().sqrt().item())
            ename = "roundtrip_l2_error"
        result["decompress"] = {"ms_per_step": round(d_ms, 4),
                                "value": round(in_bytes / d_ms / 1e6, 3), "unit": "GB/s (output)",
                                ename: err, "tolerance_abs": TOL * nrm_host,
                                "within_tolerance": bool(err <= TOL * nrm_host)}
        del back, q_work
        # the same step with the quantized values delivered as 16-bit dictionary symbols (what
        # mgh_compress feeds its Huffman stage with; an extension, never `value`: the metric
        # is defined on the reference's int64 output)
        try:
            if args.config != "512f32":
                raise mgard_amd.MgardHipError("not measured for this configuration")
            sym_out = h.decompose_quantize_sym16(d_u, mgard_amd.REL, TOL, float("inf"))
            torch.cuda.synchronize()
            t16 = time.perf_counter()
            lib, hp = mgard_amd.load_library(), h._h
            import ctypes as C
            s_sym, s_cnt, s_idx, s_val = sym_out[0], torch.zeros(1, dtype=torch.int64, device="cuda"), oidx, oval
            for _ in range(10):
                lib.mgh_decompose_quantize_sym16(hp, C.c_void_p(d_u.data_ptr()), mgard_amd.REL, TOL, float("inf"),
                                                 0.0, None, 8192, C.c_void_p(s_sym.data_ptr()),
                                                 C.c_void_p(s_cnt.data_ptr()), C.c_void_p(s_idx.data_ptr()),
                                                 C.c_void_p(s_val.data_ptr()), int(s_idx.numel()), None)
            torch.cuda.synchronize()
            s_ms = (time.perf_counter() - t16) / 10 * 1e3
            same = bool(torch.equal(s_sym.to(torch.int64), q))
            result["sym16_output"] = {"ms_per_step": round(s_ms, 4), "value": round(in_bytes / s_ms / 1e6, 3),
                                      "unit": "GB/s (input)", "equals_int64_output": same}
            del sym_out, s_sym
        except mgard_amd.MgardHipError as e:
            result["sym16_output"] = {"error": str(e)}
        # end to end through the container (reported separately, never
        # `value`): mgh_compress / mgh_decompress on the device-resident volume -- norm,
        # decompose, quantize, Huffman, serialisation into the MGARD-X container and back
        from mgard_amd import highlevel
        try:
            hl_coords = None if coords is None else [np.asarray(c, np.float64) for c in coords]
            # (output buffers allocated once: a fresh multi-GB allocation per call costs more than
            # the compression)
            obuf = torch.empty(in_bytes + 1000000, dtype=torch.uint8, device=dev)
            stream = highlevel.compress(d_u, TOL, S, mgard_amd.REL, coords=hl_coords, out=obuf)
            stream = highlevel.compress(d_u, TOL, S, mgard_amd.REL, coords=hl_coords, out=obuf)
            torch.cuda.synchronize()
            NE = 10
            # three batches of NE calls each, the best batch reported and all three listed: these
            # calls are host-latency bound (two host round trips per record), and one evidence run of
            # round 5 had a single batch at 2.4 ms between runs at 1.30-1.33 ms on either side
            c_batches, x_batches = [], []
            for _b in range(3):
                t2 = time.perf_counter()
                for _ in range(NE):
                    stream = highlevel.compress(d_u, TOL, S, mgard_amd.REL, coords=hl_coords, out=obuf)
                torch.cuda.synchronize()
                c_batches.append((time.perf_counter() - t2) / NE * 1e3)
            c_ms = min(c_batches)
            out = torch.empty_like(d_u)
            highlevel.decompress(stream, out=out)
            torch.cuda.synchronize()
            for _b in range(3):
                t3 = time.perf_counter()
                for _ in range(NE):
                    highlevel.decompress(stream, out=out)
                torch.cuda.synchronize()
                x_batches.append((time.perf_counter() - t3) / NE * 1e3)
            x_ms = min(x_batches)
            e2e_err = float((out - d_u).abs().max().item()) if S == float("inf") else \
                float(((out - d_u).double() ** 2).mean().sqrt().item())
            result["end_to_end"] = {
                "what": "mgh_compress / mgh_decompress: device-resident in, MGARD-X container "
                        "(Huffman) out, and back",
                "compress_ms": round(c_ms, 3), "compress_GBps": round(in_bytes / c_ms / 1e6, 2),
                "decompress_ms": round(x_ms, 3), "decompress_GBps": round(in_bytes / x_ms / 1e6, 2),
                "compression_ratio": round(in_bytes / int(stream.numel()), 3),
                "compress_ms_batches": [round(v, 3) for v in c_batches],
                "decompress_ms_batches": [round(v, 3) for v in x_batches],
                "within_tolerance": bool(e2e_err <= TOL * nrm_host)}
            del out, stream, obuf
            highlevel.release_cache()
        except mgard_amd.MgardHipError as e:  # the headline metric does not depend on this path
            result["end_to_end"] = {"error": str(e)}
        # ... and with HOST buffers on both sides, the way the reference is called
        if rank == 0 and dist is None and not args.no_host_leg:
            try:
                result["end_to_end_host"] = host_leg(torch, mgard_amd, u, TOL, S, nrm_host)
            except (mgard_amd.MgardHipError, RuntimeError, ValueError) as e:
                result["end_to_end_host"] = {"error": str(e)[:300]}
    q_host = None
    # ---- the other BASELINE.json configurations, on the driver's line (step-only legs; `value`
    # stays configs[1]) ----
    if args.config == "512f32" and not args.shape and not args.only_step and not args.no_other_configs:
        q_host = q.cpu().numpy() if (rank == 0 and dist is None and not args.no_cpu_baseline) else None
        del q, oidx, oval, bufs, d_u
        h.close()
        torch.cuda.empty_cache()
        oc = {}
        if dist is None:
            for name, e2e in (("512f64nu", False), ("4d", False), ("1024f32", True)):
                try:
                    oc[name] = config_leg(torch, mgard_amd, name, dev, local_rank, end_to_end=e2e,
                                          host_e2e=e2e and not args.no_host_leg)
                except (mgard_amd.MgardHipError, RuntimeError, AssertionError) as e:
                    oc[name] = {"error": str(e)[:300]}
                torch.cuda.empty_cache()
            try:
                oc["4d_volume"] = volume_leg(torch, mgard_amd, dev)
            except (mgard_amd.MgardHipError, RuntimeError, AssertionError) as e:
                oc["4d_volume"] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()
            try:
                oc["beyond_configs"] = beyond_leg(torch, mgard_amd, dev, local_rank)
            except Exception as e:  # noqa: BLE001 -- informational leg: whatever happens, the line goes out
                oc["beyond_configs"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.empty_cache()
        else:
            # N > 1: configs[3] as it is meant -- the 64 x 512^3 volume split on dim 0, one
            # 8 x 512^3 slab per rank (weak scaling), scalar norm all-reduce over RCCL
            try:
                oc["4d"] = config_leg(torch, mgard_amd, "4d", dev, local_rank, dist=dist, world=world, rank=rank)
            except (mgard_amd.MgardHipError, RuntimeError, AssertionError) as e:
                # (a failure every rank sees alike, e.g. the outlier assertion; rank-local ones are agreed
                # on inside the leg before any collective)
                oc["4d"] = {"error": str(e)[:300]}
            # ... and with the block scatter / payload gather north_star names, timed phase by phase
            try:
                oc["scatter_gather"] = scatter_gather_leg(torch, mgard_amd, dist, dev, local_rank, rank, world,
                                                          slab_t=args.sg_slab)
            except (mgard_amd.MgardHipError, RuntimeError, AssertionError) as e:
                oc["scatter_gather"] = {"error": str(e)[:300]}
        result["other_configs"] = oc
    if dist is not None:
        seen = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        result["rccl_ranks_seen"] = int(seen.item())
    if rank == 0 and dist is None and not args.no_cpu_baseline and args.config not in ("1024f32", "4d"):
        base, rq = cpu_baseline(u, TOL, S, coords)
        result["cpu_baseline"] = base
        # parity spot check on the bench workload itself: quantized integers are bit-exact
        # (s = inf only: with an L2 norm the two norms differ in the last bits by nature --
        # sequential vs tree sum -- and so may the integers; tests inject the norm instead)
        if S == float("inf"):
            qh = q_host if q_host is not None else q.cpu().numpy()
            result["parity_vs_cpu"] = bool(np.array_equal(qh, rq))
    else:
        result["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which would otherwise be flushed at exit,
        # AFTER the result: push everything out first so that the JSON line is the last line
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
