"""Second survey: hipRTC plugins and RK4InfiniteIntegrator on non-LQ problems at BL-like sizes (python scripts/api_survey2_time.py)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g
ocs = g.load_package()
from user_problems import LOGISTIC2_SRC, LOGISTIC_ROWS_CC_SRC, PREDPREY_SRC, PREDPREY_PARAMS
dev = torch.device('cuda:0')
def timeit(name, fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{name:100s} {(time.perf_counter() - t0) / reps * 1e6:10.1f} us", flush=True)
N, batch = 1000, 4096
tspan = np.linspace(0, 10, N + 1)
probs = {"registry Logistic2": ocs.LogisticProblem([3.0, 2.5], 1.5, 0.05, [[0.0, 1.0]]),
         "plugin Logistic2, full-vector methods": ocs.UserProblem(LOGISTIC2_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], [[0.0, 1.0]], has_control_char=True),
         "plugin Logistic2, row functions": ocs.UserProblem(LOGISTIC_ROWS_CC_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], [[0.0, 1.0]], has_control_char=True, row_separable=True),
         "plugin predator-prey (coupled)": ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, [[0.0, 1.0]])}
for name, prob in probs.items():
    integ = ocs.RK4Integrator(tspan)
    x0 = torch.ones((2, batch), dtype=torch.float64, device=dev)
    u = 0.05 + 0.3 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
    x = torch.empty((N + 1, 3, batch), dtype=torch.float64, device=dev); lam = torch.empty_like(x); d = torch.empty_like(u)
    def pair():
        integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
    timeit(f"{name}: pass pair, batch {batch}", pair)
    for kind, nB in (("PWLinearControl", 101), ("ChebyshevControl", 16)):
        cc = getattr(ocs, kind)(integ.t, nB, 1)
        V = torch.full((nB, batch), 0.3 if kind != "ChebyshevControl" else 0.0, dtype=torch.float64, device=dev)
        if kind == "ChebyshevControl": V[0] = 0.3
        J = torch.empty(batch, dtype=torch.float64, device=dev); dv = torch.empty_like(V)
        timeit(f"{name}: nlp_objective_dev {kind}({nB})", lambda: ocs.nlp_objective_dev(integ, prob, cc, x0, V, J=J, dJdv=dv))
    gi = ocs.RK4InfiniteIntegrator(tspan, np.linspace(10, 20, N + 1), np.array([0.3]))
    def pair_inf():
        gi.compute_states_dev(prob, x0, u, x); gi.compute_adjoints_dev(prob, u, None, lam, d)
    try:
        timeit(f"{name}: RK4InfiniteIntegrator pass pair (1000 + 1000 steps)", pair_inf)
    except Exception as e:
        print(f"{name}: RK4InfiniteIntegrator:", type(e).__name__, e)
    yg = torch.tensor(np.tile(np.array([[1.5], [1.5], [0.5], [0.5], [0.3]]), (1, batch)), dtype=torch.float64, device=dev)
    lb = torch.tensor([0.0, 0.0, -1e300, -1e300, 0.0], dtype=torch.float64, device=dev); ub = torch.tensor([1e300, 1e300, 1e300, 1e300, 1.0], dtype=torch.float64, device=dev)
    try:
        timeit(f"{name}: compute_equilibrium_dev batch {batch}", lambda: ocs.compute_equilibrium_dev(prob, yg, lb, ub, 0.05), reps=3, warm=1)
    except Exception as e:
        print(f"{name}: compute_equilibrium_dev:", type(e).__name__, e)
