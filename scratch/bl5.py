"""BASELINE config 5 (build-defined LQ32, RK4InfiniteIntegrator) through the hipRTC user-problem path:
functional check at nS = 32, nC = 4 plus a timing.  The plugin runs on the lane-per-trajectory VALU
kernels (no MFMA): this is a coverage run, not a tuned one."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
from oracle import oracle as orc
from tests.user_problems import lq_source, lq_matrices
nS, nC = 32, 4
rng = np.random.default_rng(20260405)
A = -np.diag(np.logspace(0, 3, nS)) + 0.1 * rng.normal(size=(nS, nS))   # SURVEY BL-5: stiffness ratio 1e3
Bu = rng.normal(size=(nS, nC)); q = rng.uniform(0.5, 1.5, nS); rd = rng.uniform(1, 2, nC); r = 0.05
par = np.concatenate([[r], A.ravel(order="F"), Bu.ravel(order="F"), q, rd])
bounds = [[-1.0, 1.0]] * nC
t0 = time.time()
pu = ocs.UserProblem(lq_source(nS, nC), nS, nC, par, bounds)
print(f"hipRTC build of LQ32: {time.time()-t0:.1f} s", flush=True)
T, N = 1.0, 512                     # h = T/N: |lambda_max| h = 1000/512 < 2.5 (RK4 stability, SURVEY BL-5)
tspan, tx = orc.linspace(0, T, N + 1), orc.linspace(T, 2 * T, N + 1)
batch = 1024
u = rng.uniform(-1, 1, (nC, 2 * N + 1, batch)); x0 = rng.normal(size=(nS, batch))
gi = ocs.RK4InfiniteIntegrator(tspan, tx, np.zeros(nC))
dev = torch.device("cuda:0")
x0d = torch.tensor(x0, device=dev); ud = torch.tensor(np.ascontiguousarray(u.transpose(1, 0, 2)), device=dev)
xd = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev); lamd = torch.empty_like(xd); dd = torch.empty_like(ud)
_, Jd = gi.compute_states_dev(pu, x0d, ud, xd); gi.compute_adjoints_dev(pu, ud, None, lamd, dd); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    gi.compute_states_dev(pu, x0d, ud, xd, Jd); gi.compute_adjoints_dev(pu, ud, None, lamd, dd)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"LQ32 batch={batch} N={N}+{N} tail: {dt*1e3:.2f} ms per pass pair, {batch*N/dt:.3e} steps/s", flush=True)
po = orc.LQProblem(A, Bu, q, rd, r, bounds); go = orc.RK4InfiniteIntegrator(tspan, tx, np.zeros(nC))
xh, lamh, dh, Jh = xd.cpu().numpy(), lamd.cpu().numpy(), dd.cpu().numpy(), Jd.cpu().numpy()
worst = 0.0
for b in (0, 511, 1023):
    xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b]); lamo, do = go.compute_adjoints(po, u[:, :, b])
    for a_, b_ in ((xh[:, :, b].T, xo), (lamh[:, :, b].T, lamo), (dh[:, :, b].T, do), (Jh[b], Jo)):
        worst = max(worst, float(np.max(np.abs(np.asarray(a_) - np.asarray(b_)) / np.maximum(1, np.abs(b_)))))
print(f"LQ32 max rel err vs oracle (x, lam, dJdu, J on 3 trajectories): {worst:.2e}", flush=True)
