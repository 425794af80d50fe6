"""compute_u / compute_dJdv of the three bases on device arrays (Control/*.m: u = reshape(v,nC,[]) * B, dJdv = dJdu * B'), per call."""
import os, sys, time, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
N = 1000
t = ocs.RK4Integrator(np.linspace(0, 10, N + 1)).t
def timeit(name, fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{name:70s} {(time.perf_counter() - t0) / reps * 1e6:9.1f} us", flush=True)
for batch in (4096, 16384, 65536):
    for kind, nB in (("ChebyshevControl", 16), ("ChebyshevControl", 32), ("PWLinearControl", 101), ("PWConstantControl", 50)):
        cc = getattr(ocs, kind)(t, nB, 1)
        V = torch.rand((nB, 1, batch), dtype=torch.float64, device=dev)
        u = torch.empty((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
        d = torch.rand_like(u); dv = torch.empty_like(V)
        timeit(f"batch {batch} {kind}({nB}) compute_u_dev", lambda: cc.compute_u_dev(V, u))
        timeit(f"batch {batch} {kind}({nB}) compute_dJdv_dev", lambda: cc.compute_dJdv_dev(d, dv))
