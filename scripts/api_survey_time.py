"""Per-call times of the entry points around the hot path at BL-like sizes, to spot anomalies (python scripts/api_survey_time.py)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
def timeit(name, fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{name:90s} {(time.perf_counter() - t0) / reps * 1e6:10.1f} us", flush=True)
N = 1000
tspan = np.linspace(0, 10, N + 1)
integ = ocs.RK4Integrator(tspan)
for nS, batch in ((1, 4096), (4, 4096), (4, 16384), (3, 4096)):
    prob = ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5][:nS], 1.5, 0.05, [[0.0, 1.0]])
    x0 = torch.ones((nS, batch), dtype=torch.float64, device=dev)
    for kind, nB in (("PWLinearControl", 101), ("PWConstantControl", 50), ("ChebyshevControl", 16)):
        cc = getattr(ocs, kind)(integ.t, nB, 1)
        V = torch.full((nB, batch), 0.3 if kind != "ChebyshevControl" else 0.0, dtype=torch.float64, device=dev)
        if kind == "ChebyshevControl": V[0] = 0.3
        J = torch.empty(batch, dtype=torch.float64, device=dev); dv = torch.empty_like(V)
        timeit(f"nlp_objective_dev nS={nS} batch={batch} {kind}({nB})", lambda: ocs.nlp_objective_dev(integ, prob, cc, x0, V, J=J, dJdv=dv))
    ss = torch.rand((N + 1, nS, batch), dtype=torch.float64, device=dev)
    for m in ("pchip", "linear", "previous", "nearest"):
        f = ocs.vectorInterpolant_dev(tspan, ss, m)
        tq = np.linspace(0, 10, 2001)
        timeit(f"vectorInterpolant_dev {m} nS={nS} batch={batch} 2001 points", lambda: f(tq), reps=5)
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
for batch in (1024, 65536):
    yg = torch.tensor(np.tile(np.array([[2.7], [2.2], [0.7]]), (1, batch)), device=dev)
    lb, ub = torch.tensor([0.0, -np.inf, 0.0], device=dev), torch.tensor([np.inf, np.inf, 1.0], device=dev)
    try:
        timeit(f"compute_equilibrium_dev batch={batch}", lambda: ocs.compute_equilibrium_dev(prob, yg, lb, ub, 0.05), reps=3, warm=1)
    except Exception as e:
        print("compute_equilibrium_dev:", e)
for batch in (256, 4096):
    x0 = np.ones((1, batch))
    t0 = time.perf_counter()
    try:
        r = ocs.single_shooting_batch(prob, x0, tspan, 21, MaxIter=50)
        torch.cuda.synchronize()
        print(f"{'single_shooting_batch batch=%d, 21 control points, up to 50 iterations' % batch:90s} {(time.perf_counter() - t0) * 1e3:10.1f} ms   iterations {int(r['iterations'].max())}", flush=True)
    except Exception as e:
        print("single_shooting_batch:", type(e).__name__, e)
