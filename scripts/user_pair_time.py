"""Pass pair of a hipRTC plugin given as full-vector methods (two coupled-form logistic states, tests/user_problems.LOGISTIC2_SRC) by
batch and mapping:  BATCH=... MAPPING=auto|lane|pipeline|scan python scripts/user_pair_time.py"""
import os, sys, time, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g
ocs = g.load_package()
from user_problems import LOGISTIC2_SRC
dev = torch.device('cuda:0')
batch, N = int(os.environ.get("BATCH", "4096")), 1000
NS = int(os.environ.get("NS", "2"))
if NS == 2:
    prob = ocs.UserProblem(LOGISTIC2_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], [[0.0, 1.0]], has_control_char=True)
else:   # NS logistic states from symbols, kept as full-vector methods
    import importlib, sympy as sp
    sym = importlib.import_module("ocs_amd.symbolic")
    names = ["c", "r"] + [f"m{k + 1}" for k in range(NS)]
    t, xs, lam_, us, p = sym.symbols(NS, 1, names)
    gg = sp.exp(-p["r"] * t) * (sum(xi ** 2 for xi in xs) + p["c"] * us[0] ** 2)
    ff = [xs[k] * (p[f"m{k + 1}"] - xs[k]) - us[0] for k in range(NS)]
    vals = {"c": 1.5, "r": 0.05, **{f"m{k + 1}": [3.0, 2.5, 2.0, 1.5][k] for k in range(NS)}}
    prob = ocs.make_from_symbolic(gg, ff, NS, 1, vals, [[0.0, 1.0]], allow_rows=False)
integ = ocs.RK4Integrator(np.linspace(0, 10, N + 1)).set_mapping(os.environ.get("MAPPING", "auto"))
x0 = torch.ones((NS, batch), dtype=torch.float64, device=dev)
u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
x = torch.empty((N + 1, NS + 1, batch), dtype=torch.float64, device=dev)
lam = torch.empty_like(x); d = torch.empty_like(u)
def loop(what, K=30):
    for _ in range(5):
        integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        if 'f' in what: integ.compute_states_dev(prob, x0, u, x)
        if 'b' in what: integ.compute_adjoints_dev(prob, u, None, lam, d)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
print(" ".join(f"{w} {loop(w):.1f} us" for w in ('f', 'b', 'fb', 'f', 'b')), flush=True)
