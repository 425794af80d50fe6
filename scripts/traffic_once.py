"""The kernels whose HBM traffic bench.py reports, in one short process that rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE) wraps
-- bench.py starts it twice as a child BEFORE it touches the GPU itself and reads the counter CSV (bench.py: live_traffic).
Launch sequence (the parser relies on it):
  1. ocs_copy_dev x 3            calibration: n x 8 bytes in, n x 8 bytes out per launch (8 B per lane), n x 8 = 1.05 GB
  2. BL-2 pass pair x 8          one buffer set, as the timed loop of bench.py re-uses it
  3. BL-2 pass pair x 9          three buffer sets round-robin (nothing is re-used from one step to the next)
  4. fb_sweep solve x 2          BL-3 (batch 16384): the two kernels of a sweep, live launches
Prints the byte count of the calibration launch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

ocs = g.load_package()
lib = sys.modules["ocs_amd._lib"].lib
dev = torch.device("cuda:0")
n = 1001 * 65536 * 2
src = torch.rand(n, dtype=torch.float64, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    lib.ocs_copy_dev(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_long(n), None)
torch.cuda.synchronize()
del src, dst
sys.path.insert(0, ROOT)
import bench

batch = int(os.environ.get("OCS_TRAFFIC_BATCH", str(bench.BATCH)))
tspan, x0, sets, J = bench._pair_buffers(ocs, dev, bench.NS, batch, 20260401, 3)
prob = ocs.LogisticProblem(bench.M, bench.C_PAR, bench.R_PAR, [[0.0, 1.0]])
integ = ocs.RK4Integrator(tspan)
for k in [0] * 8 + [0, 1, 2] * 3:
    u, x, lam, dJdu = sets[k]
    integ.compute_states_dev(prob, x0, u, x, J)
    integ.compute_adjoints_dev(prob, u, None, lam, dJdu)
torch.cuda.synchronize()
del sets
if os.environ.get("OCS_TRAFFIC_FBS", "1") == "1":
    rng = np.random.default_rng(20260402)
    B3 = 16384
    xs = torch.tensor(rng.uniform(0.5, 2.5, (1, B3)), device=dev)
    p3 = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
    p3.set_batch_params([0], rng.uniform(1.0, 2.0, B3)[None, :])
    g3 = ocs.RK4Integrator(ocs.linspace(0.0, bench.T_END, bench.NSTEPS + 1))
    r = ocs.fb_sweep_dev(p3, g3, xs)
    r = ocs.fb_sweep_dev(p3, g3, xs, out=r)
    torch.cuda.synchronize()
print("calibration bytes per launch (read = write):", n * 8)
