"""fb_sweep solve time at BL-3 (batch 16384): python scripts/fbs_time.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
rng = np.random.default_rng(20260402)
batch = int(os.environ.get("BATCH", "16384"))
x0 = torch.tensor(rng.uniform(0.5, 2.5, (int(os.environ.get("NS", "1")), batch)), device=dev)
cs = rng.uniform(1.0, 2.0, batch)
NS = int(os.environ.get("NS", "1"))
prob = ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5][:NS], 1.5, 0.05, [[0.0, 1.0]])
prob.set_batch_params([0], cs[None, :])
integ = ocs.RK4Integrator(ocs.linspace(0, 10, 1001))
OPTS = {"fused_update_off": int(os.environ["FUO"])} if "FUO" in os.environ else None
for _ in range(3):
    r = ocs.fb_sweep_dev(prob, integ, x0, OPTS)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        r = ocs.fb_sweep_dev(prob, integ, x0, OPTS)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    sw = r["sweeps"].cpu().numpy()
    print(f"solve {dt*1e3:.3f} ms, max sweeps {sw.max()}, batch sweeps/s {sw.max()/dt:.0f}, J[0] {float(r['J'][0]):.12f}", flush=True)
