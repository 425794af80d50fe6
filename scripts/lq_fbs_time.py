"""fb_sweep on the LQ problem (its hipRTC plugin twin: full-vector methods, lane kernels) by state count: python scripts/lq_fbs_time.py"""
import os, sys, time, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
N, batch = int(os.environ.get("N", "400")), int(os.environ.get("BATCH", "1024"))
for nS, nC in ((4, 2), (8, 2), (16, 4), (32, 4)):
    rng = np.random.default_rng(nS)
    A = -np.diag(np.linspace(0.5, 3.0, nS)) + 0.1 * rng.normal(size=(nS, nS))
    Bu = rng.normal(size=(nS, nC)); q, rd = rng.uniform(0.5, 1.5, nS), rng.uniform(1, 2, nC)
    prob = ocs.LQProblem(A, Bu, q, rd, 0.05, [[-1.0, 1.0]] * nC)
    integ = ocs.RK4Integrator(ocs.linspace(0, 2, N + 1))
    x0 = torch.tensor(rng.normal(size=(nS, batch)), device=dev)
    opts = {"nERROR_PTS": N + 1, "nINTERP_PTS": 41, "nSWEEPS": 30, "uRelax": 0.5}
    t0 = time.perf_counter(); r = ocs.fb_sweep_dev(prob, integ, x0, opts); torch.cuda.synchronize(); first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(2): r = ocs.fb_sweep_dev(prob, integ, x0, opts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    sw = r["sweeps"].cpu().numpy(); ns = max(int(sw.max()), 1) if (sw > 0).any() else 30
    print(f"LQ nS={nS} nC={nC} batch={batch} N={N}: solve {dt*1e3:.2f} ms, sweeps {sw.min()}..{sw.max()}, ~{dt/ns*1e6:.0f} us per sweep, first call (hipRTC) {first:.1f} s, path {ocs.fb_sweep_path(integ)}", flush=True)
