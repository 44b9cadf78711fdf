"""N > 1 control flow on CPU: two gloo ranks, one subdomain each; the scalar norm all-reduce and
the per-subdomain ABS tolerance reproduce the single-domain REL behaviour (the compute itself
is done by the CPU oracle here -- the distributed logic is what is under test)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.util import smooth_field


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, s, out):
    import torch.distributed as dist
    import oracle
    from mgard_amd import distributed as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle.set_num_threads(1)
    shape = (24, 17, 20)
    u = smooth_field(shape, np.float64)
    lo, hi = mdist.split_slowest(shape, world, rank)
    sub = np.ascontiguousarray(u[lo:hi])
    sval = np.float64(s)
    local = oracle.norm(sub, sval, normalize_coordinates=False)
    g = mdist.global_norm(local, s, u.size, True)
    tol = 1e-3
    atol = mdist.local_abs_tol(mdist.REL, g, tol, s, world)
    h = oracle.Hierarchy(sub.shape, np.float64)
    c = h.decompose(sub)
    q, oi, ov, n = h.quantize(c, oracle.ABS, np.float64(atol), sval, np.float64(1))
    back = h.recompose(h.dequantize(q, oracle.ABS, np.float64(atol), sval, np.float64(1),
                                    outlier_idx=oi, outlier_val=ov))
    err_inf = float(np.max(np.abs(back - sub)))
    err_sq = float(np.sum((back - sub) ** 2))
    out.put((rank, g, atol, err_inf, err_sq, sub.size))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("s", [float("inf"), 0.0])
def test_two_rank_norm_exchange_and_error_bound(s):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, s, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    u = smooth_field((24, 17, 20), np.float64)
    tol = 1e-3
    if np.isinf(s):
        ref = float(np.max(np.abs(u)))
    else:
        ref = float(np.sqrt(np.mean(u ** 2)))
    for rank, g, atol, e_inf, e_sq, n in res:
        assert abs(g - ref) <= 1e-12 * ref          # both ranks hold the GLOBAL norm
    if np.isinf(s):
        assert all(r[3] <= tol * ref for r in res)  # L-inf bound holds on every subdomain
    else:
        tot = sum(r[4] for r in res)
        assert np.sqrt(tot / u.size) <= tol * ref   # global L2 bound from the local budgets


def test_split_slowest():
    from mgard_amd import distributed as mdist
    assert [mdist.split_slowest((10, 3), 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert mdist.local_abs_tol(mdist.REL, 2.0, 1e-3, float("inf"), 8) == 2e-3
    assert mdist.local_abs_tol(mdist.ABS, 2.0, 1e-3, float("inf"), 8) == 1e-3


def _sg_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from mgard_amd import distributed as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape = (11, 6, 7)
    full = torch.arange(11 * 6 * 7, dtype=torch.float32).reshape(shape) if rank == 0 else None
    slab = mdist.scatter_slabs(full, shape, src=0, dtype=torch.float32)
    lo, hi = mdist.split_slowest(shape, world, rank)
    ok = torch.equal(slab, torch.arange(11 * 6 * 7, dtype=torch.float32).reshape(shape)[lo:hi])
    payload = torch.full((5 + 3 * rank,), rank + 1, dtype=torch.uint8)
    got = mdist.gather_payloads(payload, dst=0)
    framed = mdist.frame_payloads([g.numpy().tobytes() for g in got]) if rank == 0 else b""
    out.put((rank, ok, framed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_scatter_and_payload_gather():
    import struct
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sg_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r) for r in (out.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == struct.pack("<Q", 5) + b"\x01" * 5 + struct.pack("<Q", 8) + b"\x02" * 8


def _subgroup_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from mgard_amd import distributed as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group(ranks=[1, 2])     # group rank 0 = global rank 1, group rank 1 = global rank 2
    ok, framed = True, b""
    if rank in (1, 2):
        shape = (9, 4, 5)
        ref = torch.arange(9 * 4 * 5, dtype=torch.float32).reshape(shape)
        full = ref if rank == 1 else None
        slab = mdist.scatter_slabs(full, shape, src=0, group=grp, dtype=torch.float32)
        gr = dist.get_rank(grp)
        lo, hi = mdist.split_slowest(shape, 2, gr)
        ok = torch.equal(slab, ref[lo:hi])
        payload = torch.full((4 + 2 * gr,), gr + 7, dtype=torch.uint8)
        got = mdist.gather_payloads(payload, dst=0, group=grp)
        if gr == 0:
            framed = mdist.frame_payloads([g.numpy().tobytes() for g in got])
    out.put((rank, ok, framed))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_gather_inside_a_subgroup():
    """src / dst are ranks inside the group; the point-to-point calls need global ranks."""
    import struct
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r) for r in (out.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r][1] for r in range(world))
    assert res[1][2] == struct.pack("<Q", 4) + b"\x07" * 4 + struct.pack("<Q", 6) + b"\x08" * 6


def test_bench_launcher_starts_the_ranks_itself():
    """`python3 bench.py --gpus N` (what the driver runs) with no WORLD_SIZE in the environment must
    start N ranks itself -- fresh child processes under torch.distributed.run, never a re-exec of a
    process that has touched a GPU -- and relay rank 0's JSON line. GPU-less dry run: gloo ranks,
    the scalar norm exchange of the N > 1 path (MAX all-reduce) and the rank count."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-dry-run",
                        "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["rccl_ranks_seen"] == 2 and r["norm_exchange_ok"] is True
    assert r["steps"] == 3 and r["dry_run"] is True


def _flow_worker(rank, world, port, out):
    """The control flow of bench.py's `scatter_gather` leg at toy size on gloo: rank 0 owns the
    array, block scatter, scalar norm exchange + local ABS tolerance, one record per rank (here:
    the oracle's quantized integers of the slab as bytes -- the GPU leg puts mgh_compress there),
    payload gather, ONE container assembled on rank 0."""
    import torch
    import torch.distributed as dist
    import oracle
    from mgard_amd import distributed as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle.set_num_threads(1)
    shape = (18, 9, 10)
    u = smooth_field(shape, np.float64)
    full = torch.from_numpy(u.copy()) if rank == 0 else None
    slab = mdist.scatter_slabs(full, shape, src=0, dtype=torch.float64)
    lo, hi = mdist.split_slowest(shape, world, rank)
    same_view = rank != 0 or slab.data_ptr() == full[lo:hi].data_ptr()   # rank 0 keeps a view, no copy
    sub = np.ascontiguousarray(slab.numpy())
    g = mdist.global_norm(oracle.norm(sub, np.float64(np.inf), normalize_coordinates=False), float("inf"), u.size)
    atol = mdist.local_abs_tol(mdist.REL, g, 1e-3, float("inf"), world)
    h = oracle.Hierarchy(sub.shape, np.float64)
    # (plain integers, no dictionary: the record then needs no outlier list)
    q, oi, ov, n = h.quantize(h.decompose(sub), oracle.ABS, np.float64(atol), np.float64(np.inf), np.float64(1),
                              prep_huffman=False)
    record = torch.from_numpy(np.frombuffer(q.tobytes(), dtype=np.uint8).copy())
    payloads = mdist.gather_payloads(record, dst=0)
    ok, info = True, None
    if rank == 0:
        header = b"HEADER-OF-THE-WHOLE-DOMAIN"
        cont = mdist.assemble_container(header, payloads).numpy().tobytes()
        ok = cont == header + mdist.frame_payloads([p.numpy().tobytes() for p in payloads])
        # the records decode to slabs that meet the GLOBAL relative bound
        at = len(header)
        errs = []
        for r in range(world):
            size = int(np.frombuffer(cont[at:at + 8], dtype="<u8")[0])
            a, b = mdist.split_slowest(shape, world, r)
            qq = np.frombuffer(cont[at + 8:at + 8 + size], dtype=np.int64).reshape((b - a,) + shape[1:])
            hr = oracle.Hierarchy(qq.shape, np.float64)
            back = hr.recompose(hr.dequantize(qq.copy(), oracle.ABS, np.float64(atol), np.float64(np.inf), np.float64(1),
                                              prep_huffman=False))
            errs.append(float(np.max(np.abs(back - u[a:b]))))
            at += 8 + size
        ok = ok and at == len(cont)
        info = (max(errs), 1e-3 * g)
    out.put((rank, bool(ok and same_view), info, g))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_compress_gather_flow_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flow_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r) for r in (out.get(timeout=180) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][3] == res[1][3] == float(np.max(np.abs(smooth_field((18, 9, 10), np.float64))))
    err, bound = res[0][2]
    assert err <= bound
