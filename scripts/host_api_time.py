"""PCIe-inclusive rate of the host-buffer entry points (ocs_integrator_compute_states / _adjoints with host arrays in the
reference's layouts) at BL-2, next to the device-resident rate bench.py reports:  python scripts/host_api_time.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
B, N, NS = 4096, 1000, 4
rng = np.random.default_rng(20260401)
prob = ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5], 1.5, 0.05, [[0.0, 1.0]])
integ = ocs.RK4Integrator(ocs.linspace(0.0, 10.0, N + 1))
x0 = rng.uniform(0.8, 2.0, (NS, B))
u = np.asfortranarray(rng.uniform(0.05, 0.45, (1, 2 * N + 1, B)))   # the reference's (column-major) layout: no conversion
x0 = np.asfortranarray(x0)
for rep in range(3):
    t0 = time.perf_counter()
    x, J = integ.compute_states(prob, x0, u)
    lam, dJdu = integ.compute_adjoints(prob, u)
    dt = time.perf_counter() - t0
    moved = u.nbytes * 2 + x.nbytes + lam.nbytes + dJdu.nbytes
    print(f"host API pass pair: {dt*1e3:.1f} ms = {B*N/dt:.3e} steps/s, {moved/1e6:.0f} MB over PCIe ({moved/dt/1e9:.1f} GB/s incl. layout changes)", flush=True)

# the same through the C entry points with caller-owned, already-touched output arrays (what a host program that reuses
# its buffers sees: no first-touch page faults of fresh allocations)
import ctypes as C
from importlib import import_module
lib = ocs._lib.lib if hasattr(ocs, "_lib") else None
if lib is not None:
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    x0f = np.asfortranarray(x0); uf = np.asfortranarray(u.reshape(1, 2 * N + 1, B))
    x = np.zeros((NS + 1, N + 1, B), order="F"); J = np.zeros(B)
    lam = np.zeros((NS + 1, N + 1, B), order="F"); dJdu = np.zeros((1, 2 * N + 1, B), order="F")
    for rep in range(3):
        t0 = time.perf_counter()
        lib.ocs_compute_states(integ._h, prob._h, B, P(x0f), P(uf), P(x), P(J))
        t1 = time.perf_counter()
        lib.ocs_compute_adjoints(integ._h, prob._h, B, P(uf), None, P(lam), P(dJdu))
        dt = time.perf_counter() - t0
        print(f"C entry points, reused buffers: states {1e3*(t1-t0):.1f} ms + adjoints {1e3*(dt-(t1-t0)):.1f} ms = {B*N/dt:.3e} steps/s", flush=True)
