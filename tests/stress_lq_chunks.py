"""Stress of the time-parallel LQ passes (csrc/ocs_lq_kernels.hip, "chunked passes"): random shapes -- nS 1..32, nC 1..4, N 2..700,
batch 1..300 (ragged groups of 16), non-uniform grids, with and without a tail leg, explicit lamT, partial outputs -- the requested
chunked mapping (4) against the serial one-wave mapping (1) of the same library, 1e-11 relative.  python tests/stress_lq_chunks.py
[ncases] (on the GPU box); tests/test_gpu_lq.py runs a short version."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def run(ocs, ncases=40, seed=2026, verbose=True):
    from tests.user_problems import lq_matrices
    rng = np.random.default_rng(seed)
    worst, failed = 0.0, 0
    for case in range(ncases):
        nS, nC = int(rng.integers(1, 33)), int(rng.integers(1, 5))
        N = int(rng.choice([2, 3, 5, 8, 17, 64, 100, 257, 700]))
        batch = int(rng.choice([1, 5, 16, 17, 48, 100, 300]))
        tail = bool(rng.integers(0, 2))
        A, Bu, q, rdiag = lq_matrices(nS, nC, int(rng.integers(1, 10**6)))
        tspan = np.concatenate([[0.0], np.sort(rng.uniform(0.0, 2.0, N - 1)), [2.0]])
        tspan = 0.5 * (tspan + np.linspace(0.0, 2.0, N + 1))
        hmax = np.diff(tspan).max()
        if tail:
            N2 = int(rng.choice([2, 9, 40, 130]))
            tx, ustar = np.linspace(2.0, 3.0, N2 + 1), rng.uniform(-0.3, 0.3, nC)
            hmax = max(hmax, 1.0 / N2)
        # a stable system inside RK4's stability region on BOTH grids: with an unstable A, or |lambda| h > 2.8 anywhere, the
        # costate grows by many orders of magnitude and the serial and the chunked passes, which round differently, differ by
        # that growth times round-off (1e-9 at a growth of 1e10 was what an earlier version of this generator produced)
        ev = np.linalg.eigvals(A)
        A = A - max(0.0, ev.real.max() + 0.2) * np.eye(nS)
        A = A / max(1.0, np.abs(np.linalg.eigvals(A)).max() * hmax / 2.5)
        prob = ocs.LQProblem(A, Bu, q, rdiag, 0.05, [[-1.0, 1.0]] * nC)
        u = rng.uniform(-1, 1, (nC, 2 * N + 1, batch))
        x0 = rng.normal(size=(nS, batch))
        if tail:
            mk = lambda: ocs.RK4InfiniteIntegrator(tspan, tx, ustar)
        else:
            mk = lambda: ocs.RK4Integrator(tspan)
        gc, gs = mk().set_mapping(4), mk().set_mapping(1)
        lamT = None if tail or rng.integers(0, 2) else rng.normal(size=(nS + 1, batch))
        xc, Jc = gc.compute_states(prob, x0, u)
        lc, dc = gc.compute_adjoints(prob, u, lamT) if lamT is not None else gc.compute_adjoints(prob, u)
        xs, Js = gs.compute_states(prob, x0, u)
        ls, ds = gs.compute_adjoints(prob, u, lamT) if lamT is not None else gs.compute_adjoints(prob, u)
        parts = {"x": relerr(xc, xs), "J": relerr(Jc, Js), "lam": relerr(lc, ls), "dJdu": relerr(dc, ds)}
        e = max(parts.values())
        ok = e < 1e-11 and np.array_equal(Jc, xc[-1, -1, :] if not tail else Jc) and np.all(np.isfinite(dc))
        worst = max(worst, e)
        failed += not ok
        if verbose or not ok:
            print(f"case {case:3d}: nS={nS:2d} nC={nC} N={N:3d} batch={batch:3d} tail={int(tail)} lamT={int(lamT is not None)} "
                  f"err {e:.2e} {'ok' if ok else 'FAILED'}" + ("" if ok else f" {parts} N2={N2 if tail else 0} "
                  f"lam err by column (first 6 / last 3): {[float(f'{relerr(lc[:, k], ls[:, k]):.1e}') for k in (0, 1, 2, 3, 4, 5, N - 2, N - 1, N)]} "
                  f"|lam|max {np.abs(ls).max():.2e} |x|max {np.abs(xs).max():.2e}"), flush=True)
    print(f"worst error {worst:.2e}; failed cases: {failed}", flush=True)
    return failed, worst


if __name__ == "__main__":
    import __graft_entry__ as g
    f, _ = run(g.load_package(), int(sys.argv[1]) if len(sys.argv) > 1 else 40)
    sys.exit(1 if f else 0)
