"""Multi-GPU host logic of the hot path: one process per GPU, one subdomain per rank.

The decomposition/quantization of a subdomain is independent of every other subdomain
(reference include/mgard-x/DomainDecomposer/DomainDecomposer.hpp:260-303); the only exchange is
the scalar reduction that turns per-subdomain norms into the global norm a REL bound refers to
(include/mgard-x/CompressionHighLevel/ErrorToleranceCalculator.hpp:69-89, 91-131), after which
every subdomain runs with an ABS tolerance (`calc_local_abs_tol`, :134-155;
CompressionHighLevel.hpp:122-144). `torch.distributed` with backend "nccl" is RCCL over xGMI on
MI355X; the same code runs on "gloo" for the CPU tests.
"""
import math

REL, ABS = 0, 1


def global_norm(local_norm, s, total_num_elem, normalize_coordinates=True, group=None,
                device=None):
    """Combine subdomain norms (each computed WITHOUT coordinate normalisation, as
    calc_norm_decomposed does) into the norm of the whole domain with one scalar all-reduce:
    MAX for s = inf, SUM of squares otherwise."""
    import torch
    import torch.distributed as dist
    if math.isinf(s):
        t = torch.tensor([float(local_norm)], dtype=torch.float64, device=device)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t.item())
    t = torch.tensor([float(local_norm) ** 2], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    v = float(t.item())
    return math.sqrt(v / total_num_elem) if normalize_coordinates else math.sqrt(v)


def local_abs_tol(ebtype, norm, tol, s, num_subdomains):
    """calc_local_abs_tol (ErrorToleranceCalculator.hpp:134-155)."""
    if ebtype == REL:
        if math.isinf(s):
            return tol * norm
        return math.sqrt((tol * norm) * (tol * norm) / num_subdomains)
    if math.isinf(s):
        return tol
    return math.sqrt((tol * tol) / num_subdomains)


def split_slowest(shape, world_size, rank):
    """Contiguous slabs along the slowest dimension (domain_decomposition_dim = 0, sizes as equal
    as possible). Returns (start, stop) of this rank's slab."""
    n = shape[0]
    base, rem = divmod(n, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def scatter_slabs(full, shape, src=0, group=None, device=None, dtype=None, copy=False):
    """Block scatter of the domain decomposition (reference: DomainDecomposer::copy_subdomain,
    DomainDecomposer.hpp:649-845, across devices): rank `src` holds the whole array `full`
    (shape `shape`, C order) and sends every other rank its contiguous slab along the slowest
    dimension with point-to-point sends (RCCL send/recv over xGMI with backend "nccl"; there is
    no halo, subdomains share no nodes). Returns this rank's slab. `full` is only read on `src`;
    the other ranks pass None and must give `dtype`.
    ALIASING: on `src` the returned slab is a VIEW of `full` (nothing is copied there): changing or
    freeing `full` afterwards changes the slab. Pass copy=True for a slab of its own. `full` must be
    C-contiguous (ValueError otherwise)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return full
    # `src` and the loop index are ranks INSIDE `group`; isend / recv address peers by GLOBAL rank
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    g = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    lo, hi = split_slowest(shape, world, rank)
    if rank == src:
        # slabs of the slowest dimension are contiguous views: every send streams straight out of
        # `full` (no staging copy), all of them in flight at once -- one xGMI link per peer
        if not full.is_contiguous():
            raise ValueError("scatter_slabs: the array to scatter must be C-contiguous")
        reqs = []
        for r in range(world):
            if r == src:
                continue
            a, b = split_slowest(shape, world, r)
            reqs.append(dist.isend(full[a:b], dst=g(r), group=group))
        for q in reqs:
            q.wait()
        return full[lo:hi].clone() if copy else full[lo:hi]  # (a view unless asked otherwise)
    mine = torch.empty((hi - lo,) + tuple(shape[1:]), dtype=dtype, device=device)
    dist.recv(mine, src=g(src), group=group)
    return mine


def gather_payloads(payload, dst=0, group=None):
    """Payload gather: every rank's compressed subdomain (a 1-D uint8 tensor of its own length)
    ends up on `dst`, in subdomain-id (= rank) order, the way the reference concatenates
    `[u64 size][payload]` per subdomain (GPUPipelines.hpp:189-193). Lengths travel first (one
    small all-gather), the bytes by point-to-point sends. Returns the list of payloads on `dst`,
    None elsewhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [payload]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    g = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    if rank != dst:
        dist.send(payload.contiguous(), dst=g(dst), group=group)
        return None
    out = []
    for r in range(world):
        if r == dst:
            out.append(payload)
        else:
            buf = torch.empty(int(sizes[r].item()), dtype=payload.dtype, device=payload.device)
            dist.recv(buf, src=g(r), group=group)
            out.append(buf)
    return out


def assemble_container(header, payloads):
    """The container `mgh_decompress` opens, on the device the payloads live on: `header` (bytes:
    preamble + header of the WHOLE domain, declaring the decomposition) followed by
    `[u64 LE size][payload]` per subdomain in id order (GPUPipelines.hpp:189-193). `payloads`: 1-D
    uint8 tensors. Returns one uint8 tensor."""
    import struct
    import torch
    dev = payloads[0].device
    parts = [torch.frombuffer(bytearray(header), dtype=torch.uint8).to(dev)]
    for p in payloads:
        parts.append(torch.frombuffer(bytearray(struct.pack("<Q", int(p.numel()))), dtype=torch.uint8).to(dev))
        parts.append(p)
    return torch.cat(parts)


def frame_payloads(payloads):
    """`[u64 LE compressed_size][payload]` per subdomain, concatenated in id order
    (GPUPipelines.hpp:189-193)."""
    import struct
    return b"".join(struct.pack("<Q", len(p)) + bytes(p) for p in payloads)
