"""Batched single shooting (ocs_single_shooting_batch_dev, SPG): wall time of a solve against the GPU time of its kernels.
python scripts/ss_batch_time.py   (BATCH, NSTEPS, NCP, MAXIT in the environment)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
batch, N, ncp, maxit = (int(os.environ.get(k, d)) for k, d in (("BATCH", "4096"), ("NSTEPS", "500"), ("NCP", "101"), ("MAXIT", "40")))
rng = np.random.default_rng(1)
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
prob.set_batch_params([0], rng.uniform(1.0, 2.0, batch)[None, :])
x0 = rng.uniform(0.8, 2.0, (1, batch))
tspan = ocs.linspace(0, 10, N + 1)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ocs.single_shooting_batch(prob, x0, tspan, ncp, MaxIter=maxit)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = r["iterations"].cpu().numpy()
    print(f"solve {dt*1e3:.1f} ms, iterations {it.min()}..{it.max()}, converged {float(r['converged'].float().mean()):.2f}, "
          f"{dt/max(1,it.max())*1e6:.0f} us per outer iteration", flush=True)
