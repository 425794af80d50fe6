import sys, numpy as np, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
rng = np.random.default_rng(20260402)
batch = 16384
x0 = torch.tensor(rng.uniform(0.5, 2.5, (1, batch)), device=dev)
cs = rng.uniform(1.0, 2.0, batch)
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
prob.set_batch_params([0], cs[None, :])
integ = ocs.RK4Integrator(ocs.linspace(0, 10, 1001))
for _ in range(3):
    r = ocs.fb_sweep_dev(prob, integ, x0)
torch.cuda.synchronize()
