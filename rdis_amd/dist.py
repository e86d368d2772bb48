"""Multi-GPU plumbing: one process per GPU, independent components sharded across
ranks, no data-path collective; the only exchange is the sum of the per-rank
objective partials (the top-level sum over components of the reference,
src/RDISOptimizer.cpp:1491-1494) -- an all-reduce of one fp64 over RCCL / xGMI
(backend "nccl" on ROCm), or gloo on CPU for the tests.

Single large components (BASELINE configs 2-4) do not shard: the factor-parallel
split would need an all-reduce of the whole gradient at every line-search trial
(SURVEY.md 8e).  bench.py therefore gives every rank its own component (weak scaling).
"""
from __future__ import annotations

import numpy as np

from .problems import PackedProblem, shard_components


def rank_decomposition(pp: PackedProblem, rank: int, world: int):
    """CSR (free_ptr, free_vid, fac_ptr, fac_id) and component ids of the components this
    rank solves: LPT by factor count, identical on every rank."""
    weights = np.diff(pp.comp_fac_ptr)
    mine = shard_components(pp.ncomp, weights, world)[rank]
    free_ptr, fac_ptr = [0], [0]
    free_vid, fac_id = [], []
    for c in mine:
        fv, fc = pp.component(int(c))
        free_vid.append(fv)
        fac_id.append(fc)
        free_ptr.append(free_ptr[-1] + len(fv))
        fac_ptr.append(fac_ptr[-1] + len(fc))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
    return (np.array(free_ptr, dtype=np.int64), cat(free_vid).astype(np.int64),
            np.array(fac_ptr, dtype=np.int64), cat(fac_id).astype(np.int64), mine)


def allreduce_objective(local_sum, dist=None, device=None) -> float:
    """Sum of the per-rank objective partials.  `local_sum` is a float or a 1-element
    torch tensor (possibly a device view of the solver's objective buffer, reduced in
    place).  With dist=None (single process) it is returned unchanged."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(local_sum if not hasattr(local_sum, "item") else local_sum.item())
    import torch
    t = local_sum if hasattr(local_sum, "item") else torch.tensor([float(local_sum)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())


def gather_deterministic_sum(local_sum: float, dist=None, device=None) -> float:
    """Reproducible variant: gather the partials and add them in rank order on every
    rank (an all-reduce may associate differently run to run)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(local_sum)
    import torch
    t = torch.tensor([float(local_sum)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    s = 0.0
    for o in out:
        s += float(o.item())
    return s
