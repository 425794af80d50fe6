"""Pass pair at small batches (the reference's own use: fmincon evaluates one candidate at a time): python scripts/small_batch_time.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
for nS, N in ((1, 500), (1, 1000), (4, 1000)):
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
    integ = ocs.RK4Integrator(ocs.linspace(0, 10, N + 1))
    for batch in (1, 7, 64 // nS, 2 * 64 // nS + 1, 256):
        x0 = torch.ones((nS, batch), dtype=torch.float64, device=dev)
        u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
        x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
        lam = torch.empty_like(x); d = torch.empty_like(u)
        for _ in range(5):
            integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 30
        for _ in range(K):
            integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
        torch.cuda.synchronize()
        print(f"nS={nS} N={N} batch={batch}: {(time.perf_counter() - t0) / K * 1e6:.0f} us per pass pair", flush=True)
