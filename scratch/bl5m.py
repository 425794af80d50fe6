"""BL-5 (LQ32 + RK4InfiniteIntegrator) on the matrix-core kernels: timing sweep over batch."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
nS, nC = int(os.environ.get("NS", 32)), int(os.environ.get("NC", 4))
rng = np.random.default_rng(20260405)
A = -np.diag(np.logspace(0, 3, nS)) + 0.1 * rng.normal(size=(nS, nS))
Bu = rng.normal(size=(nS, nC)); q = rng.uniform(0.5, 1.5, nS); rd = rng.uniform(1, 2, nC)
prob = ocs.LQProblem(A, Bu, q, rd, 0.05, [[-1.0, 1.0]] * nC)
T, N = 1.0, int(os.environ.get("NSTEPS", 512))
tspan, tx = np.linspace(0, T, N + 1), np.linspace(T, 2 * T, N + 1)
dev = torch.device("cuda:0")
for batch in [int(b) for b in os.environ.get("BATCHES", "1024,8192,16384,32768").split(",")]:
    u = torch.tensor(rng.uniform(-1, 1, (2 * N + 1, nC, batch)), device=dev)
    x0 = torch.tensor(rng.normal(size=(nS, batch)), device=dev)
    gi = ocs.RK4InfiniteIntegrator(tspan, tx, np.zeros(nC)); gi.set_mapping(int(os.environ.get('MAPPING', 0)))
    xd = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev); lamd = torch.empty_like(xd); dd = torch.empty_like(u)
    _, Jd = gi.compute_states_dev(prob, x0, u, xd); gi.compute_adjoints_dev(prob, u, None, lamd, dd); torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps): gi.compute_states_dev(prob, x0, u, xd, Jd)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): gi.compute_adjoints_dev(prob, u, None, lamd, dd)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / reps
    steps = batch * 2 * N   # both legs
    fl = steps * (12 * 2 * nS * nS + 3 * 2 * 2 * nS * nC + 2 * 2 * nS * nC)
    print(f"map={os.environ.get('MAPPING', 0)} LQ{nS} nC={nC} batch={batch} N={N}+{N}: fwd {tf*1e3:.2f} ms  bwd {tb*1e3:.2f} ms  pair {1e3*(tf+tb):.2f} ms  "
          f"{steps/(tf+tb):.3e} steps/s (both legs)  {fl/(tf+tb)/1e12:.2f} TFLOP/s  finite={bool(torch.isfinite(lamd).all())}", flush=True)
