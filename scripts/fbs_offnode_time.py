"""fb_sweep with the reference's default options (1001 error points, fb_sweep.m:21) on grids whose nodes are / are not those points:
python scripts/fbs_offnode_time.py   (BATCH, NS in the environment)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
rng = np.random.default_rng(20260402)
batch, NS = int(os.environ.get("BATCH", "16384")), int(os.environ.get("NS", "1"))
x0 = torch.tensor(rng.uniform(0.5, 2.5, (NS, batch)), device=dev)
cs = rng.uniform(1.0, 2.0, batch)
prob = ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5][:NS], 1.5, 0.05, [[0.0, 1.0]])
prob.set_batch_params([0], cs[None, :])
for N, opts, label in ((1000, None, "N = 1000, defaults (error points = nodes)"), (500, None, "N = 500, defaults (1001 error points: every node and midpoint)"),
                       (504, None, "N = 504, defaults (1001 error points off the grid)"), (504, {"nERROR_PTS": 505}, "N = 504, nERROR_PTS = 505 (= nodes)"),
                       (2000, None, "N = 2000, defaults (1001 error points: every other node)")):
    integ = ocs.RK4Integrator(ocs.linspace(0, 10, N + 1))
    for _ in range(2): r = ocs.fb_sweep_dev(prob, integ, x0, opts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = ocs.fb_sweep_dev(prob, integ, x0, opts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    sw = r["sweeps"].cpu().numpy()
    print(f"{label:75s}: solve {dt*1e3:8.3f} ms, sweeps {sw.min()}..{sw.max()}, {dt/max(sw.max(),1)*1e6/N*1000:7.1f} us per sweep per 1000 steps, path {ocs.fb_sweep_path(integ)}", flush=True)
