"""A coupled plugin beyond nS = 4 / nC = 2 (six stocks on a ring, three controls; generated from symbols): pass pair and
fb_sweep timings at batch 4096 x 1000 steps.  python scripts/big_plugin_time.py"""
import os, sys, time, importlib, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g
ocs = g.load_package()
sym = importlib.import_module("ocs_amd.symbolic")
from user_problems import ring6_symbolic
dev = torch.device('cuda:0')
batch, N = int(os.environ.get("BATCH", "4096")), int(os.environ.get("N", "1000"))
gg, f, vals = ring6_symbolic(sym)
prob = ocs.make_from_symbolic(gg, f, 6, 3, vals, [[0.0, 1.0]] * 3)
integ = ocs.RK4Integrator(ocs.linspace(0, 4, N + 1))
if os.environ.get("MAPPING"):
    integ.set_mapping(int(os.environ["MAPPING"]))
gen = torch.Generator(device=dev).manual_seed(5)
x0 = torch.rand((6, batch), dtype=torch.float64, device=dev, generator=gen) * 1.2 + 0.6
u = torch.rand((2 * N + 1, 3, batch), dtype=torch.float64, device=dev, generator=gen)
x = torch.empty((N + 1, 7, batch), dtype=torch.float64, device=dev)
lam = torch.empty_like(x); dJdu = torch.empty_like(u)
_, J = integ.compute_states_dev(prob, x0, u, x)
integ.compute_adjoints_dev(prob, u, None, lam, dJdu)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for rep in range(4):
    ev[0].record()
    for _ in range(10): integ.compute_states_dev(prob, x0, u, x, J)
    ev[1].record()
    for _ in range(10): integ.compute_adjoints_dev(prob, u, None, lam, dJdu)
    ev[2].record(); torch.cuda.synchronize()
    print(f"state pass {ev[0].elapsed_time(ev[1])*100:.1f} us, adjoint pass {ev[1].elapsed_time(ev[2])*100:.1f} us; "
          f"checksums {float(J.sum()):.12e} {float(lam[0].abs().sum()):.12e} {float(dJdu.abs().sum()):.12e}", flush=True)
opts = {"nSWEEPS": 60, "uRelax": 0.35}
for _ in range(2):
    r = ocs.fb_sweep_dev(prob, integ, x0, opts)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    r = ocs.fb_sweep_dev(prob, integ, x0, opts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
sw = r["sweeps"].cpu().numpy()
print(f"fb_sweep: solve {dt*1e3:.2f} ms, sweeps {sw.min()}..{sw.max()}, {dt/max(sw.max(), 1)*1e6:.0f} us per sweep, path {r.get('path')}, J[0] {float(r['J'][0]):.12f}", flush=True)
