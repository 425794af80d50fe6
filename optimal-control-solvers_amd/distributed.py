"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Trajectories are independent, so the batch axis shards into contiguous blocks with NO data-path
exchange; grid, basis and problem parameters are replicated (KBs).  The only
collectives are O(1)-size post-reductions: the all-reduce(SUM) of [sum J, sum dJdv, count] when the
batch is an ensemble whose mean objective is optimised, and an all-gather of per-rank (min J, argmin)
for best-candidate selection (RCCL has no MINLOC).  Works with any backend (tests use gloo on CPU)."""
from __future__ import annotations

import os

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


def collectives_on():
    """Whether the post-reductions go through torch.distributed: a process group with more than one rank -- or, with
    OCS_FORCE_COLLECTIVES=1, any initialised process group (one rank: every collective, barrier and the max-over-ranks
    timing then execute as they would at N ranks; bench.py's rehearsal of the RCCL code path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("OCS_FORCE_COLLECTIVES") == "1"


def ensemble_objective(J_local: torch.Tensor, dJdv_local: torch.Tensor | None = None):
    """Mean objective (and mean gradient) of the global ensemble from each rank's shard.
    J_local [B_loc], dJdv_local [nV][B_loc].  One all-reduce of 2 + nV doubles."""
    nV = 0 if dJdv_local is None else dJdv_local.shape[0]
    buf = torch.zeros(2 + nV, dtype=torch.float64, device=J_local.device)
    buf[0] = J_local.sum()
    buf[1] = float(J_local.numel())
    if nV:
        buf[2:] = dJdv_local.sum(dim=1)
    if collectives_on():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    Jmean = buf[0] / buf[1]
    return (Jmean, None) if not nV else (Jmean, buf[2:] / buf[1])


def best_candidate(J_local: torch.Tensor, lo: int):
    """Global (min J, global index) over all shards; `lo` is this shard's first global index."""
    jmin, imin = torch.min(J_local, dim=0)
    pair = torch.stack([jmin.to(torch.float64), (imin + lo).to(torch.float64)])
    world, _ = world_info()
    if not collectives_on():
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
    if not collectives_on():
        return J_local.clone()
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros(mx, dtype=J_local.dtype, device=J_local.device)
    pad[: J_local.numel()] = J_local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: h - l] for o, (l, h) in zip(out, sizes)])


def sharded_objective_eval(eval_local, V_global: torch.Tensor, want_gradient_mean: bool = True):
    """One sharded objective+gradient evaluation of a global batch of candidates (BASELINE config 4: 65 536
    coefficient vectors over 8 GPUs): every rank evaluates its contiguous block of columns of V_global [nV][total]
    with `eval_local(V_local) -> (J_local [B_loc], dJdv_local [nV][B_loc])` -- no data-path exchange -- and the
    O(1)-size post-reductions follow: all-gather of J (gather_objectives), best candidate (best_candidate), ensemble
    mean of J and dJdv (ensemble_objective).  bench.py drives the GPU kernels through this function and
    tests/test_distributed_gloo.py the CPU oracle (gloo), so the sharding / argument path is the same code.
    Returns dict(lo, hi, J_local, dJdv_local, J_all, best=(Jmin, argmin), J_mean, dJdv_mean)."""
    world, rank = world_info()
    total = V_global.shape[1]
    lo, hi = shard_bounds(total, world, rank)
    V_local = V_global[:, lo:hi].contiguous()
    J_local, G_local = eval_local(V_local)
    out = {"lo": lo, "hi": hi, "J_local": J_local, "dJdv_local": G_local}
    out["J_all"] = gather_objectives(J_local, total)
    out["best"] = best_candidate(J_local, lo)
    Jm, Gm = ensemble_objective(J_local, G_local if want_gradient_mean else None)
    out["J_mean"], out["dJdv_mean"] = Jm, Gm
    return out


def timed_max_over_ranks(fn, reps: int, sync=None):
    """Wall time of `reps` calls of fn(), bracketed by a barrier and `sync()` on both sides, maximum over ranks
    (the contract bench.py's headline loop follows)."""
    import time
    on = collectives_on()
    if sync:
        sync()
    if on:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        sync()
    if on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if on:
        t = torch.tensor([dt], dtype=torch.float64, device=_reduce_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt / reps


def prewarm(fn, seconds: float, sync=None, chunk: int = 1):
    """Untimed calls of fn() for at least `seconds` of wall time (an idle GPU sits at low clocks for a few hundred ms;
    a count-based warm-up of a ms-scale leg does not bring them up).  Returns the number of calls."""
    import time
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(chunk):
            fn()
        n += chunk
        if sync:
            sync()
        if time.perf_counter() - t0 >= seconds:
            return n


def timed_reps_max_over_ranks(fn, reps: int, sync=None, inner: int = 1):
    """Per-repetition wall times (seconds per call of fn): every repetition is `inner` calls bracketed by sync() on both
    sides; under a process group a barrier precedes the first repetition and every repetition's time is the maximum over
    ranks.  The caller reports min / median / max and the list, so that one stalled repetition shows as what it is."""
    import time
    on = collectives_on()
    if sync:
        sync()
    if on:
        dist.barrier()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        if sync:
            sync()
        ts.append((time.perf_counter() - t0) / inner)
    if on:
        t = torch.tensor(ts, dtype=torch.float64, device=_reduce_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts = [float(v) for v in t.tolist()]
    return ts


def spread(ts, unit_scale: float = 1e3, digits: int = 4):
    """min / median / max / per-rep list of a list of times (default: seconds -> ms)."""
    a = sorted(ts)
    n = len(a)
    med = a[n // 2] if n % 2 else 0.5 * (a[n // 2 - 1] + a[n // 2])
    r = lambda v: round(v * unit_scale, digits)
    return {"min": r(a[0]), "median": r(med), "max": r(a[-1]), "reps": n, "per_rep": [r(v) for v in ts]}, med


def _reduce_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
