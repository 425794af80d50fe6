"""Batched version of the reference's own test setup (solve_test_problem.m: TestOCProblem, N = 500, PWLinear with 101
points): objective + gradient per batch, fused vs unfused control basis."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device("cuda:0")
N, nP = 500, 101
integ = ocs.RK4Integrator(ocs.linspace(0.0, 10.0, N + 1))
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
for kind in ("PWLinear", "PWConstant"):
    ctrl = (ocs.PWLinearControl if kind == "PWLinear" else ocs.PWConstantControl)(integ.t, nP if kind == "PWLinear" else nP - 1, 1)
    for batch in [int(b) for b in os.environ.get("BATCHES", "4096,65536").split(",")]:
        rng = np.random.default_rng(1)
        V = rng.uniform(0.1, 0.9, (ctrl.nBasis, batch))
        vd = torch.tensor(V, device=dev); x0 = torch.ones((1, batch), dtype=torch.float64, device=dev)
        J = torch.empty(batch, dtype=torch.float64, device=dev); G = torch.empty_like(vd)
        res = {}
        for mode in ("off", "on"):
            ctrl.set_fusion(mode)
            for _ in range(2): ocs.nlp_objective_dev(integ, prob, ctrl, x0, vd, (), J, G)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): ocs.nlp_objective_dev(integ, prob, ctrl, x0, vd, (), J, G)
            torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 5, J.clone(), G.clone())
        dJ = float((res["on"][1] - res["off"][1]).abs().max()); dG = float((res["on"][2] - res["off"][2]).abs().max())
        print(f"{kind} batch={batch}: unfused {res['off'][0]*1e3:.3f} ms  fused {res['on'][0]*1e3:.3f} ms  "
              f"({batch*N/res['on'][0]:.3e} steps/s)  max|dJ|={dJ:.2e} max|dG|={dG:.2e}", flush=True)
