"""fb_sweep on a full-vector hipRTC plugin (two logistic states, tests/user_problems.LOGISTIC2_SRC) by batch: BATCH=... python scripts/fbs_vector_time.py"""
import os, sys, time, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g
ocs = g.load_package()
from user_problems import LOGISTIC2_SRC
dev = torch.device('cuda:0')
batch, N = int(os.environ.get("BATCH", "4096")), 1000
rng = np.random.default_rng(3)
x0 = torch.tensor(rng.uniform(0.8, 1.6, (2, batch)), device=dev)
integ = ocs.RK4Integrator(ocs.linspace(0, 10, N + 1))
prob = ocs.UserProblem(LOGISTIC2_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], [[0.0, 1.0]], has_control_char=True)
for _ in range(2):
    r = ocs.fb_sweep_dev(prob, integ, x0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    r = ocs.fb_sweep_dev(prob, integ, x0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
sw = r["sweeps"].cpu().numpy()
print(f"solve {dt*1e3:.2f} ms, sweeps {sw.min()}..{sw.max()}, {dt/sw.max()*1e6:.0f} us per sweep, path {ocs.fb_sweep_path(integ)}, J[0] {float(r['J'][0]):.12f}", flush=True)
