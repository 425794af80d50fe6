import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
N, nB = 1000, 16
for batch in (8192, 65536):
    rng = np.random.default_rng(20260403)
    V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]; V[0] += 0.5
    integ = ocs.RK4Integrator(np.linspace(0, 10, N + 1))
    prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
    ctrl = ocs.ChebyshevControl(integ.t, nB, 1)
    vd = torch.tensor(V, device=dev); x0 = torch.ones((1, batch), dtype=torch.float64, device=dev)
    J = torch.empty(batch, dtype=torch.float64, device=dev); G = torch.empty_like(vd)
    for _ in range(3): ocs.nlp_objective_dev(integ, prob, ctrl, x0, vd, (), J, G)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
    for _ in range(reps): ocs.nlp_objective_dev(integ, prob, ctrl, x0, vd, (), J, G)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"BL-4 batch={batch}: {dt*1e3:.3f} ms per objective+gradient evaluation of the batch, {batch*N/dt:.3e} steps/s, "
          f"alg(32 B/step) {32*batch*N/dt/1e9:.0f} GB/s")
