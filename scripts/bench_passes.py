"""Times compute_states / compute_adjoints (device entry points) per mapping and batch.
  MAPPING=auto|lane|rowsplit|pipeline|scan BATCHES=4096,16384 NSTEPS=1000 python scripts/bench_passes.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
def run(nS, N, batch, reps=10):
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
    integ = ocs.RK4Integrator(np.linspace(0, 10, N + 1)).set_mapping(os.environ.get('MAPPING','auto'))
    x0 = torch.ones((nS, batch), dtype=torch.float64, device=dev)
    u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
    x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    lam = torch.empty_like(x); d = torch.empty_like(u)
    try:
        integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
    except Exception as e:
        print(f"nS={nS} N={N} batch={batch}: unsupported ({str(e)[:60]})"); return
    integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
    torch.cuda.synchronize()
    ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(reps):
        ef[0].record(); integ.compute_states_dev(prob, x0, u, x); ef[1].record()
        integ.compute_adjoints_dev(prob, u, None, lam, d); ef[2].record()
        torch.cuda.synchronize()
        tf += ef[0].elapsed_time(ef[1]); tb += ef[1].elapsed_time(ef[2])
    tf /= reps; tb /= reps
    nA = nS + 1
    bytes_ = 8 * (3 * nA + 6) * batch * N
    print(f"nS={nS} N={N} batch={batch}: fwd {tf*1e3:.1f} us  bwd {tb*1e3:.1f} us  steps/s {batch*N/((tf+tb)*1e-3):.3e}  "
          f"alg GB/s {bytes_/((tf+tb)*1e-3)/1e9:.1f}  frac {bytes_/((tf+tb)*1e-3)/8e12:.3f}", flush=True)
import os
batches = [int(b) for b in os.environ.get("BATCHES", "4096,16384,65536,262144").split(",")]
for nS in (4, 1):
    for batch in batches:
        run(nS, int(os.environ.get('NSTEPS','1000')), batch)
