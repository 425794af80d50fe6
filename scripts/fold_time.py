"""Time per sweep of the folded fb_sweep kernels with convergence switched off (every solve runs nSWEEPS sweeps):
python scripts/fold_time.py   (OCS_LIB_OVERRIDE selects an alternative build of the library)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
rng = np.random.default_rng(20260402)
batch = int(os.environ.get("BATCH", "16384"))
x0 = torch.tensor(rng.uniform(0.5, 2.5, (1, batch)), device=dev)
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
prob.set_batch_params([0], rng.uniform(1.0, 2.0, batch)[None, :])
integ = ocs.RK4Integrator(ocs.linspace(0, 10, 1001))
NSW = 21
opts = {"nSWEEPS": NSW, "uRelTol": 1e-300, "uAbsTol": 1e-300, "fused_update_off": int(os.environ.get("FUO", "0"))}
for _ in range(2):
    ocs.fb_sweep_dev(prob, integ, x0, opts)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(3):
        ocs.fb_sweep_dev(prob, integ, x0, opts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{dt*1e6/NSW:.1f} us per sweep (solve {dt*1e3:.3f} ms)", flush=True)
