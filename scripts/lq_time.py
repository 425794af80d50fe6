"""BL-5 pass pair timing (LQ32, RK4InfiniteIntegrator 4000 + 4000 steps, batch 8192): python scripts/lq_time.py
(N=..., BATCH=... in the environment for other shapes; OCS_LIB_OVERRIDE selects another build of the library)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
nS, nC, T = 32, 4, 10.0
N, batch = int(os.environ.get("N", "4000")), int(os.environ.get("BATCH", "8192"))
rng = np.random.default_rng(20260405)
A = -np.diag(np.logspace(0, 3, nS)) + 0.1 * rng.normal(size=(nS, nS))
Bu = rng.normal(size=(nS, nC))
q, rd = rng.uniform(0.5, 1.5, nS), rng.uniform(1, 2, nC)
prob = ocs.LQProblem(A, Bu, q, rd, 0.05, [[-1.0, 1.0]] * nC)
integ = ocs.RK4InfiniteIntegrator(np.linspace(0, T, N + 1), np.linspace(T, 2 * T, N + 1), np.zeros(nC))
if os.environ.get("MAPPING"):   # 1 one wave / 2 two / 3 four waves per 16 trajectories, 4 time-parallel chunks
    integ.set_mapping(int(os.environ["MAPPING"]))
gen = torch.Generator(device=dev).manual_seed(20260405)
u = torch.rand((2 * N + 1, nC, batch), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
x0 = torch.randn((nS, batch), dtype=torch.float64, device=dev, generator=gen)
x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
lam = torch.empty_like(x); dJdu = torch.empty_like(u)
_, J = integ.compute_states_dev(prob, x0, u, x)
integ.compute_adjoints_dev(prob, u, None, lam, dJdu)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
fl_bwd = 7 * 2 * nS * nS + 2 * 2 * nS * nC + 2 * 2 * nS * nC
fl_fwd = 4 * 2 * nS * nS + 3 * 2 * nS * nC
for rep in range(int(os.environ.get("REPS", "3"))):
    ev[0].record(); integ.compute_states_dev(prob, x0, u, x, J)
    ev[1].record(); integ.compute_adjoints_dev(prob, u, None, lam, dJdu)
    ev[2].record(); torch.cuda.synchronize()
    tf, tb = ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3
    st = batch * 2 * N
    print(f"forward {tf*1e3:.2f} ms ({st*fl_fwd/tf/1e12:.1f} TF = {st*fl_fwd/tf/78.6e12:.3f}), adjoint {tb*1e3:.2f} ms "
          f"({st*fl_bwd/tb/1e12:.1f} TF = {st*fl_bwd/tb/78.6e12:.3f}); checksum {float(lam[0].abs().sum()):.12e} {float(dJdu.abs().sum()):.12e}", flush=True)
