"""world_size-2 gloo test of the N>1 path on CPU: the batch shards into contiguous blocks with no
data-path exchange, and the O(1) post-reductions (ensemble mean of J / dJdv, best candidate,
J all-gather) reproduce the single-process result.  The per-shard compute is the CPU oracle here
(no GPU in this container); on the GPU box the same helpers run over RCCL in bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    ocs = g.load_package()
    from oracle import oracle as orc
    P = {"c": 1.5, "m": 3.0, "r": 0.05}
    tspan = orc.linspace(0, 10, 101)
    rng = np.random.default_rng(7)                       # same stream on every rank: global inputs
    V = rng.uniform(0.1, 0.9, (11, total))
    lo, hi = ocs.distributed.shard_bounds(total, world, rank)
    po, go = orc.TestOCProblem(P, [[0.0, 1.0]]), orc.RK4Integrator(tspan)
    co = orc.PWLinearControl(go.t, 11, 1)
    J = np.empty(hi - lo)
    G = np.empty((11, hi - lo))
    for k, b in enumerate(range(lo, hi)):
        J[k], G[:, k], _ = orc.nlp_objective(go, po, co, [1.0], V[:, b])
    Jt, Gt = torch.tensor(J), torch.tensor(G)
    Jm, Gm = ocs.distributed.ensemble_objective(Jt, Gt)
    best = ocs.distributed.best_candidate(Jt, lo)
    allJ = ocs.distributed.gather_objectives(Jt, total)

    # the driver bench.py uses for BASELINE config 4 under world > 1 (here with the oracle as the per-shard evaluator)
    def eval_local(Vl):
        Jl = np.empty(Vl.shape[1])
        Gl = np.empty((11, Vl.shape[1]))
        for k in range(Vl.shape[1]):
            Jl[k], Gl[:, k], _ = orc.nlp_objective(go, po, co, [1.0], Vl[:, k].numpy())
        return torch.tensor(Jl), torch.tensor(Gl)
    r = ocs.distributed.sharded_objective_eval(eval_local, torch.tensor(V))
    assert (r["lo"], r["hi"]) == (lo, hi) and torch.equal(r["J_all"], allJ) and r["best"] == best
    assert abs(float(r["J_mean"]) - float(Jm)) < 1e-15 and torch.allclose(r["dJdv_mean"], Gm, rtol=0, atol=1e-15)
    dt = ocs.distributed.timed_max_over_ranks(lambda: None, 3)
    assert dt >= 0.0
    if rank == 0:
        q.put((float(Jm), Gm.numpy(), best, allJ.numpy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_batch():
    import __graft_entry__ as g
    ocs = g.load_package()
    for total in (1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            b = [ocs.distributed.shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1
    assert ocs.distributed.shard_bounds(65536, 8, 3) == (24576, 32768)  # BL-4: 8192 per GPU


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process():
    total, world = 13, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Jm, Gm, best, allJ, (lo, hi) = res
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    P = {"c": 1.5, "m": 3.0, "r": 0.05}
    tspan = orc.linspace(0, 10, 101)
    V = np.random.default_rng(7).uniform(0.1, 0.9, (11, total))
    po, go = orc.TestOCProblem(P, [[0.0, 1.0]]), orc.RK4Integrator(tspan)
    co = orc.PWLinearControl(go.t, 11, 1)
    J = np.empty(total)
    G = np.empty((11, total))
    for b in range(total):
        J[b], G[:, b], _ = orc.nlp_objective(go, po, co, [1.0], V[:, b])
    assert (lo, hi) == (0, 7)
    assert np.array_equal(allJ, J)                                   # no data-path exchange: bit-identical
    assert abs(Jm - J.mean()) < 1e-13 * abs(J.mean())
    np.testing.assert_allclose(Gm, G.mean(axis=1), rtol=1e-12, atol=1e-14)
    assert best == (float(J.min()), int(J.argmin()))
