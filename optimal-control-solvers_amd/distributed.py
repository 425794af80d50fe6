"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Trajectories are independent, so the batch axis shards into contiguous blocks with NO data-path
exchange; grid, basis and problem parameters are replicated (KBs).  The only
collectives are O(1)-size post-reductions: the all-reduce(SUM) of [sum J, sum dJdv, count] when the
batch is an ensemble whose mean objective is optimised, and an all-gather of per-rank (min J, argmin)
for best-candidate selection (RCCL has no MINLOC).  Works with any backend (tests use gloo on CPU)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous block [lo, hi) of a batch of `total` for `rank` of `world`; remainders go to the
    first ranks so sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def ensemble_objective(J_local: torch.Tensor, dJdv_local: torch.Tensor | None = None):
    """Mean objective (and mean gradient) of the global ensemble from each rank's shard.
    J_local [B_loc], dJdv_local [nV][B_loc].  One all-reduce of 2 + nV doubles."""
    nV = 0 if dJdv_local is None else dJdv_local.shape[0]
    buf = torch.zeros(2 + nV, dtype=torch.float64, device=J_local.device)
    buf[0] = J_local.sum()
    buf[1] = float(J_local.numel())
    if nV:
        buf[2:] = dJdv_local.sum(dim=1)
    world, _ = world_info()
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    Jmean = buf[0] / buf[1]
    return (Jmean, None) if not nV else (Jmean, buf[2:] / buf[1])


def best_candidate(J_local: torch.Tensor, lo: int):
    """Global (min J, global index) over all shards; `lo` is this shard's first global index."""
    jmin, imin = torch.min(J_local, dim=0)
    pair = torch.stack([jmin.to(torch.float64), (imin + lo).to(torch.float64)])
    world, _ = world_info()
    if world == 1:
        return float(pair[0]), int(pair[1])
    gathered = [torch.empty_like(pair) for _ in range(world)]
    dist.all_gather(gathered, pair)
    allp = torch.stack(gathered)
    k = int(torch.argmin(allp[:, 0]))
    return float(allp[k, 0]), int(allp[k, 1])


def gather_objectives(J_local: torch.Tensor, total: int):
    """All-gather of the full J vector (BL-4: 64 KB per rank).  Shards may differ by one element, so
    each rank pads to the largest shard."""
    world, rank = world_info()
    if world == 1:
        return J_local.clone()
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros(mx, dtype=J_local.dtype, device=J_local.device)
    pad[: J_local.numel()] = J_local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: h - l] for o, (l, h) in zip(out, sizes)])
