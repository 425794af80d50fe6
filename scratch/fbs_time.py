import sys, numpy as np, torch, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
rng = np.random.default_rng(20260402)
batch = int(os.environ.get("BATCH", 16384))
x0 = torch.tensor(rng.uniform(0.5, 2.5, (1, batch)), device=dev)
cs = rng.uniform(1.0, 2.0, batch)
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
prob.set_batch_params([0], cs[None, :])
integ = ocs.RK4Integrator(ocs.linspace(0, 10, 1001))
opts = {"fused_update_off": int(os.environ.get("OFF", 0)), "nWINDOWS": int(os.environ.get("NWIN", 0))}
r = ocs.fb_sweep_dev(prob, integ, x0, opts); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): r = ocs.fb_sweep_dev(prob, integ, x0, opts)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
sw = r["sweeps"].cpu().numpy()
print(f"batch={batch} off={opts['fused_update_off']}: {dt*1e3:.3f} ms per solve, {sw.max()} sweeps -> {dt/max(sw.max(),1)*1e6:.1f} us per sweep", flush=True)
