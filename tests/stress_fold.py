"""Randomised comparison of the folded fb_sweep kernels against the unfolded path (fused_update_off = 3) and the oracle,
PER INSTANCE (functions/fb_sweep.m:79-87, 99-115):

  * both paths and the oracle must agree on the number of sweeps of every instance (0 = not converged: the reference
    returns an empty struct, fb_sweep.m:77 -- nothing else is compared for such an instance);
  * a converged instance must agree fold-vs-unfolded to 1e-11 and with the oracle to 1e-10 (x, lam, u on the interpolation
    points, J; relative to max(1, |ref|)) whatever the other instances of its batch do.

Library for tests/test_gpu_fb_sweep.py::test_fold_stress (fixed seed); as a script: python tests/stress_fold.py [ncases]
(SEED, ORACLE_ALL=1 to check every converged instance against the oracle instead of a sample)."""
import os
import sys

import numpy as np

P = {"c": 1.5, "m": 3.0, "r": 0.05}


def _rel(a, b):
    """max |a - b| / max(1, |b|); where b is not finite (a trajectory that ran off on a coarse grid) a must match exactly"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    bad = ~np.isfinite(b)
    if bad.any() and not np.all((np.isnan(a[bad]) & np.isnan(b[bad])) | (a[bad] == b[bad])):
        return float("inf")
    ok = ~bad
    if not ok.any():
        return 0.0
    with np.errstate(invalid="ignore"):
        e = np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok]))
    return float("inf") if np.isnan(e).any() else float(e.max())


def draw_case(rng, case, ocs, ragged=False):
    nS = int(rng.choice([1, 2, 4]))
    N = 8 * int(rng.integers(1, 40))
    batch = (64 // nS) * int(rng.integers(1, 5))
    T = float(rng.uniform(0.5, 6.0))
    kind = rng.integers(0, 3) if case % 4 == 3 else rng.integers(0, 2)   # (an irregular grid takes the unfolded path)
    if kind == 0:
        tspan = ocs.linspace(0, T, N + 1) if case % 3 else np.linspace(0, T, N + 1)   # MATLAB-style or numpy linspace
    elif kind == 1:
        tspan = np.arange(N + 1) * 2.0 ** -5
    else:
        tspan = np.sort(np.concatenate([[0.0, T], rng.uniform(0, T, N - 1)]))
        if np.min(np.diff(tspan)) < 1e-4 * T / N:
            tspan = np.linspace(0, T, N + 1)
    lbv = float(rng.choice([0.0, 0.1, -0.2]))
    ubv = lbv + float(rng.uniform(0.3, 1.2))
    if case % 2:   # every other case: an upper bound the control does not reach
        ubv = 6.0
    c = dict(nS=nS, N=N, batch=batch, kind=int(kind), tspan=tspan, bounds=[[lbv, ubv]], m=[3.0, 2.5, 2.0, 1.5][:nS],
             x0=rng.uniform(0.8, 1.6, (nS, batch)), cs=rng.uniform(1.0, 2.0, batch), cost_row=int(rng.integers(0, 2)))
    if ragged and case % 3 == 1:
        # a ragged last tile (taken by a workgroup that overlaps its neighbour): an even number of further instances, from a
        # generator of their own (the cases of the pytest list keep their draws)
        r2 = np.random.default_rng(1000 + case)
        extra = 2 * int(r2.integers(1, 64 // nS // 2))
        c["x0"] = np.hstack([c["x0"], r2.uniform(0.8, 1.6, (nS, extra))])
        c["cs"] = np.concatenate([c["cs"], r2.uniform(1.0, 2.0, extra)])
        c["batch"] = batch + extra
    return c


def run_case(ocs, oracle, c, nsweeps=60, oracle_instances=None):
    """-> dict(ok, sweeps, err_fold, err_oracle, worst: description of the worst converged instance, failures: list)"""
    prob = ocs.LogisticProblem(c["m"], P["c"], P["r"], c["bounds"])
    prob.set_batch_params([0], c["cs"][None, :])
    base = {"nERROR_PTS": c["N"] + 1, "nINTERP_PTS": 17, "nSWEEPS": nsweeps, "cost_row": c["cost_row"]}
    ra = ocs.fb_sweep_batch(prob, c["x0"], c["tspan"], dict(base))
    rd = ocs.fb_sweep_batch(prob, c["x0"], c["tspan"], dict(base, fused_update_off=3))
    sw = ra["sweeps"]
    fails = []
    if not np.array_equal(sw, rd["sweeps"]):
        fails.append(f"sweeps differ fold/unfolded at instances {np.nonzero(sw != rd['sweeps'])[0][:8].tolist()}")
    conv = np.nonzero((sw > 0) & (rd["sweeps"] > 0))[0]
    err_fold, worst = 0.0, None
    for b in conv:
        e = max(_rel(ra[k][..., b], rd[k][..., b]) for k in ("x", "lam", "u"))
        e = max(e, _rel(ra["J"][b:b + 1], rd["J"][b:b + 1]))
        k_ = int(sw[b])
        e_mc = _rel(ra["maxChange"][:k_, b], rd["maxChange"][:k_, b])
        if not (e < 1e-11 and e_mc < 1e-6):
            fails.append(f"instance {b} ({k_} sweeps): fold-vs-unfolded {e:.2e}, maxChange {e_mc:.2e}")
        if e >= err_fold:
            err_fold, worst = e, f"instance {b} ({k_} sweeps)"
    # the oracle: the slowest-converging instances, the ends of the batch and a not-converged one
    if oracle_instances is None:
        order = conv[np.argsort(-sw[conv])] if conv.size else conv
        pick = set(order[:3].tolist()) | {0, c["batch"] - 1} | set(np.nonzero(sw == 0)[0][:1].tolist())
        oracle_instances = sorted(pick)
    err_or = 0.0
    for b in oracle_instances:
        ref = oracle.fb_sweep(oracle.LogisticProblem(c["m"], c["cs"][b], P["r"], c["bounds"]), c["x0"][:, b], c["tspan"], base)
        if ref["_sweeps"] != sw[b]:
            fails.append(f"instance {b}: {sw[b]} sweeps, oracle {ref['_sweeps']}")
            continue
        if ref["_sweeps"] == 0:
            continue
        e = max(_rel(ra["J"][b:b + 1], np.array([ref["J"]])), _rel(ra["lam"][:, :, b], ref["lam"]),
                _rel(ra["x"][:c["nS"], :, b], ref["x"]), _rel(ra["u"][:, :, b], ref["u"]))
        err_or = max(err_or, e)
        if not e < 1e-10:
            fails.append(f"instance {b} ({sw[b]} sweeps): vs oracle {e:.2e}")
    return dict(ok=not fails, sweeps=sw, err_fold=err_fold, err_oracle=err_or, worst=worst, failures=fails,
                checked=len(oracle_instances))


if __name__ == "__main__":
    ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    ocs = g.load_package()
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    bad = 0
    for case in range(ncases):
        c = draw_case(rng, case, ocs, ragged=os.environ.get("RAGGED", "1") != "0")
        inst = list(range(c["batch"])) if os.environ.get("ORACLE_ALL") else None
        r = run_case(ocs, oracle, c, oracle_instances=inst)
        sw = r["sweeps"]
        print(f"case {case}: nS={c['nS']} N={c['N']} batch={c['batch']} grid={c['kind']} lb={c['bounds'][0][0]} "
              f"sweeps {sw.min()}..{sw.max()} ({int((sw == 0).sum())} not converged) fold-vs-unfolded {r['err_fold']:.2e} "
              f"[{r['worst']}] vs-oracle {r['err_oracle']:.2e} ({r['checked']} instances) {'ok' if r['ok'] else 'FAIL'}", flush=True)
        for f in r["failures"][:6]:
            print("     ", f)
        bad += not r["ok"]
    print("failed cases:", bad)
    sys.exit(1 if bad else 0)
