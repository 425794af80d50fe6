"""Times the shooting objective + gradient (BASELINE configs[3]: TestOCProblem, Chebyshev-16, N = 1000) per fusion mode.
  BATCHES=8192,65536 MODES=on,lane NB=16 python scripts/bl4_time.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
N, nB = int(os.environ.get('NSTEPS', '1000')), int(os.environ.get('NB', '16'))
integ = ocs.RK4Integrator(np.linspace(0, 10, N + 1))
NS = int(os.environ.get("NS", "0"))   # 0: TestOCProblem (BL-4); 1..4: LogisticK with that many states
prob = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]]) if NS == 0 else ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5][:NS], 1.5, 0.05, [[0.0, 1.0]])
cc = ocs.ChebyshevControl(integ.t, nB, 1)
for batch in [int(b) for b in os.environ.get('BATCHES', '64,8192,65536').split(',')]:
    rng = np.random.default_rng(20260403)
    V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.5
    Vd = torch.tensor(V, device=dev)
    x0 = torch.ones((max(NS, 1), batch), dtype=torch.float64, device=dev)
    res = {}
    for mode in os.environ.get('MODES', 'on,lane').split(','):
        cc.set_fusion(mode)
        for _ in range(5):
            J, dv = ocs.nlp_objective_dev(integ, prob, cc, x0, Vd)
        torch.cuda.synchronize()
        reps = int(os.environ.get('REPS', '20'))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            J, dv = ocs.nlp_objective_dev(integ, prob, cc, x0, Vd)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[mode] = (J.cpu().numpy(), dv.cpu().numpy())
        print(f"batch {batch} nB {nB} mode {mode}: {ms*1e3:.1f} us per evaluation, {batch*N/(ms*1e-3):.3e} steps/s, "
              f"16 B/step: {16*batch*N/(ms*1e-3)/1e12:.2f} TB/s", flush=True)
    if os.environ.get('DUMP'):
        for k, (Jv, dv_) in res.items():
            np.savez(f"{os.environ['DUMP']}_{batch}_{k}.npz", J=Jv, dJdv=dv_)
    ks = list(res)
    for k in ks[1:]:
        a, b = res[ks[0]], res[k]
        print(f"   {ks[0]} vs {k}: J {np.max(np.abs(a[0]-b[0])/np.maximum(1,np.abs(b[0]))):.2e}  dJdv {np.max(np.abs(a[1]-b[1])/np.maximum(1,np.abs(b[1]))):.2e}")
