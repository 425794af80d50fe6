"""Steady-state loop timings: K forwards back to back, K adjoints back to back, K pairs (MAPPING=auto|pipeline|scan|lane,
OCS_FWD_V1=1 for the previous state kernel).  The pair is what bench.py times; the difference to the sum of the two
single-kernel loops is the interference between the kernels (cache write-back, clocks)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
nS, N, batch = int(os.environ.get('NS', '4')), int(os.environ.get('NSTEPS', '1000')), int(os.environ.get('BATCH', '4096'))
m = [3.0, 2.5, 2.0, 1.5][:nS]
prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
# UNIFORM=1: steps of 2^-7 (bitwise uniform: the kernels keep the step sizes in registers); default MATLAB-style linspace
tspan = np.arange(N + 1) / 128.0 if os.environ.get('UNIFORM') else np.linspace(0, 10, N + 1)
integ = ocs.RK4Integrator(tspan).set_mapping(os.environ.get('MAPPING', 'auto'))
x0 = torch.ones((nS, batch), dtype=torch.float64, device=dev)
u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
lam = torch.empty_like(x); d = torch.empty_like(u)
def loop(what, K=50):
    for _ in range(5):
        integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        if 'f' in what: integ.compute_states_dev(prob, x0, u, x)
        if 'b' in what: integ.compute_adjoints_dev(prob, u, None, lam, d)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
for what in ('f', 'b', 'fb', 'f', 'b', 'fb'):
    print(what, f"{loop(what):.1f} us per iteration", flush=True)
