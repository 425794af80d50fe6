"""Pass pair (compute_states + compute_adjoints, full output) of a coupled user problem (predator-prey, hipRTC) against the
row-separable registry problem with the same shapes, per mapping.  BATCHES=4096 NSTEPS=1000 python scripts/vector_time.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
from tests.user_problems import PREDPREY_PARAMS, PREDPREY_SRC, PREDPREY_TC_SRC
ocs = g.load_package()
dev = torch.device('cuda:0')
N = int(os.environ.get('NSTEPS', '1000'))
tspan = np.linspace(0, 6, N + 1)
probs = {"predator-prey (user, coupled)": ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, [[0.0, 1.0]]),
         "predator-prey (user, coupled, ocs_tcoef)": ocs.UserProblem(PREDPREY_TC_SRC, 2, 1, PREDPREY_PARAMS, [[0.0, 1.0]]),
         "Logistic2 (registry, row-separable)": ocs.LogisticProblem([3.0, 2.5], 1.5, 0.05, [[0.0, 1.0]])}
for batch in [int(b) for b in os.environ.get('BATCHES', '512,4096,16384').split(',')]:
    x0 = 1.0 + torch.rand((2, batch), dtype=torch.float64, device=dev)
    u = torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev) * 0.5
    x = torch.empty((N + 1, 3, batch), dtype=torch.float64, device=dev)
    lam, d = torch.empty_like(x), torch.empty_like(u)
    for name, prob in probs.items():
        for mapping in ("auto", "lane"):
            integ = ocs.RK4Integrator(tspan).set_mapping(mapping)
            for _ in range(5):
                integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            reps = 10
            for _ in range(reps):
                ev[0].record(); integ.compute_states_dev(prob, x0, u, x); ev[1].record()
                integ.compute_adjoints_dev(prob, u, None, lam, d); ev[2].record()
                torch.cuda.synchronize()
                tf += ev[0].elapsed_time(ev[1]) / reps; tb += ev[1].elapsed_time(ev[2]) / reps
            print(f"batch {batch} {name} mapping {mapping}: fwd {tf*1e3:.1f} us  bwd {tb*1e3:.1f} us  pair {(tf+tb)*1e3:.1f} us", flush=True)
