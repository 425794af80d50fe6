"""BL-2 pass pair (compute_states + compute_adjoints, full output) over ROTATE complete buffer sets used round-robin
(ROTATE=1: the headline loop's single set, whose u and x can stay in the 256 MiB memory-side cache between steps;
ROTATE=3: 1.4 GB in flight, nothing survives from one step to the next).  Prints us per pair and, from HIP events, per
kernel.  Environment: ROTATE, BATCH, NS, K (pairs per timed loop), WHAT=fb|f|b.
Under rocprofv3 --kernel-trace --stats the per-kernel averages of the two modes tell which pass pays for the misses."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
nS, N = int(os.environ.get('NS', '4')), 1000
batch, R, K = int(os.environ.get('BATCH', '4096')), int(os.environ.get('ROTATE', '3')), int(os.environ.get('K', '120'))
what = os.environ.get('WHAT', 'fb')
m = [3.0, 2.5, 2.0, 1.5][:nS]
prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
integ = ocs.RK4Integrator(np.linspace(0, 10, N + 1)).set_mapping(os.environ.get('MAPPING', 'auto'))
x0 = torch.ones((nS, batch), dtype=torch.float64, device=dev)
sets = []
for k in range(R):
    u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
    x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    sets.append((u, x, torch.empty_like(x), torch.empty_like(u)))
J = torch.empty(batch, dtype=torch.float64, device=dev)
def step(k, ev=None):
    u, x, lam, d = sets[k % R]
    if ev: ev[0].record()
    if 'f' in what: integ.compute_states_dev(prob, x0, u, x, J)
    if ev: ev[1].record()
    if 'b' in what: integ.compute_adjoints_dev(prob, u, None, lam, d)
    if ev: ev[2].record()
for k in range(R):  # every set gets a valid x (WHAT=b reads it)
    u, x, lam, d = sets[k]
    integ.compute_states_dev(prob, x0, u, x, J)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for k in range(30): step(k)
    torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K): step(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e6
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
    for k in range(30): step(k, evs[k])
    torch.cuda.synchronize()
    f = np.median([e[0].elapsed_time(e[1]) for e in evs]) * 1e3
    b = np.median([e[1].elapsed_time(e[2]) for e in evs]) * 1e3
    print(f"ROTATE={R} batch={batch} nS={nS} {what}: {dt:.1f} us per step   events: forward {f:.1f} us, adjoint {b:.1f} us", flush=True)
