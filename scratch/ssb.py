import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
rng = np.random.default_rng(3)
B, N, nPts = 10, 100, 11
m = [3.0, 2.0]
tspan = ocs.linspace(0, 5, N + 1)
x0 = np.vstack([rng.uniform(0.8, 1.5, B), rng.uniform(0.8, 1.5, B)])
cs = rng.uniform(1.0, 2.0, B)
prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
prob.set_batch_params([0], cs[None, :])
for tolx in (1e-5, 1e-9, 1e-13):
    r = ocs.single_shooting_batch(prob, x0, tspan, nPts, u0=0.3, TolFun=1e-7, TolX=tolx, MaxIter=600, FreeInitStates=[2], FreeStateBounds=[[0.5, 1.2]])
    print(tolx, r["iterations"].cpu().numpy(), r["projected_gradient"].cpu().numpy(), r["J"].cpu().numpy()[:3], r["v"].cpu().numpy()[:, 0])
from scipy.optimize import minimize
integ = ocs.RK4Integrator(tspan); ctl = ocs.PWLinearControl(integ.t, nPts, 1)
p1 = ocs.LogisticProblem(m, cs[0], 0.05, [[0.0, 1.0]])
def f(v):
    J, g, _ = ocs.nlp_objective(integ, p1, ctl, x0[:, :1].copy(), v[:, None], FreeInitStates=[2])
    return float(J[0]), g[:, 0]
res = minimize(f, np.concatenate([np.full(nPts, 0.3), [x0[1, 0]]]), jac=True, method="SLSQP", bounds=[(0, 1)] * nPts + [(0.5, 1.2)], options={"ftol": 1e-13, "maxiter": 500})
print("slsqp", res.fun, res.x, res.nit)
print("grad at slsqp opt", f(res.x)[1])
r = ocs.single_shooting_batch(prob, x0, tspan, nPts, u0=0.3, TolFun=1e-7, MaxIter=600)
print("nofree", r["iterations"].cpu().numpy(), r["projected_gradient"].cpu().numpy())
