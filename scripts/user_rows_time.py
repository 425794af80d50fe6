"""Pass timings of a user problem given as row functions (hipRTC instances of k_forward_p2 / k_backward_scan) against the
registry problem: python scripts/user_rows_time.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
from tests.user_problems import LOGISTIC_ROWS_SRC
dev = torch.device('cuda:0')
nS, N, B = 4, 1000, 4096
m = [3.0, 2.5, 2.0, 1.5]
pb = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
pu = ocs.UserProblem(LOGISTIC_ROWS_SRC, nS, 1, [1.5, 0.05] + m, [[0.0, 1.0]], row_separable=True)
x0 = torch.ones((nS, B), dtype=torch.float64, device=dev)
u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, B), dtype=torch.float64, device=dev)
for name, prob in (("registry", pb), ("rows", pu), ("registry", pb), ("rows", pu)):
    integ = ocs.RK4Integrator(np.linspace(0, 10, N + 1))
    x = torch.empty((N + 1, nS + 1, B), dtype=torch.float64, device=dev)
    lam, d = torch.empty_like(x), torch.empty_like(u)
    def loop(what, K=50):
        for _ in range(5):
            integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            if 'f' in what: integ.compute_states_dev(prob, x0, u, x)
            if 'b' in what: integ.compute_adjoints_dev(prob, u, None, lam, d)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
    loop('fb', 200)
    print(name, " ".join(f"{w} {loop(w):.1f} us" for w in ('f', 'b', 'fb')), flush=True)
