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
