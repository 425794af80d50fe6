"""Host time of one ocs_multi_*_dev call against the one-device _dev call it wraps (device-resident blocks, one device):
the calls only enqueue, so a loop of K calls without synchronisation measures the host cost per call as long as the
kernels are shorter than it -- a batch of 64 trajectories x 8 steps keeps them out of the way.
python scripts/multi_overhead.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
N, B, K = 8, 64, 2000
tspan = np.linspace(0, 1, N + 1)
md = ocs.MultiDevice([0])
integs = md.replicate(lambda: ocs.RK4Integrator(tspan))
probs = md.replicate(lambda: ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5], 1.5, 0.05, [[0.0, 1.0]]))
g1, p1 = ocs.RK4Integrator(tspan), ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5], 1.5, 0.05, [[0.0, 1.0]])
x0 = torch.ones((4, B), dtype=torch.float64, device=dev)
u = 0.2 * torch.ones((2 * N + 1, 1, B), dtype=torch.float64, device=dev)
x = torch.empty((N + 1, 5, B), dtype=torch.float64, device=dev); J = torch.empty(B, dtype=torch.float64, device=dev)
lam, d = torch.empty_like(x), torch.empty_like(u)
import ctypes as C
lib = ocs._lib.lib
ga, pa = (C.c_void_p * 1)(integs[0]._h), (C.c_void_p * 1)(probs[0]._h)
cnt = (C.c_int * 1)(B)
P = lambda t: (C.c_void_p * 1)(t.data_ptr())
ax0, au, ax, aJ, al, ad = P(x0), P(u), P(x), P(J), P(lam), P(d)
def multi_pair():
    lib.ocs_multi_compute_states_dev(md._h, ga, pa, cnt, ax0, au, ax, aJ, 0)
    lib.ocs_multi_compute_adjoints_dev(md._h, ga, pa, cnt, au, None, al, ad)
def single_pair():
    lib.ocs_compute_states_dev(g1._h, p1._h, B, x0.data_ptr(), u.data_ptr(), x.data_ptr(), J.data_ptr(), None)
    lib.ocs_compute_adjoints_dev(g1._h, p1._h, B, u.data_ptr(), None, lam.data_ptr(), d.data_ptr(), None)
def loop(fn):
    for _ in range(50): fn()
    torch.cuda.synchronize(); md.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    dt = (time.perf_counter() - t0) / K * 1e6
    torch.cuda.synchronize(); md.synchronize()
    return dt
for rep in range(3):
    a, b = loop(single_pair), loop(multi_pair)
    print(f"host time per pass pair of calls: one-device _dev {a:.1f} us, ocs_multi_*_dev (1 device) {b:.1f} us -> {0.5*(b-a):.1f} us per call over the one-device call", flush=True)
def multi_pair_red():
    lib.ocs_multi_compute_states_dev(md._h, ga, pa, cnt, ax0, au, ax, aJ, 1)
    lib.ocs_multi_compute_adjoints_dev(md._h, ga, pa, cnt, au, None, al, ad)
print(f"with the RCCL reductions enqueued behind the state pass: {loop(multi_pair_red):.1f} us per pair of calls", flush=True)
if os.environ.get("OCS_MULTI_ALLOW_DUPLICATES") == "1":
    for n in (2, 4):
        md2 = ocs.MultiDevice([0] * n)
        ig = md2.replicate(lambda: ocs.RK4Integrator(tspan)); pb = md2.replicate(lambda: ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5], 1.5, 0.05, [[0.0, 1.0]]))
        gg, pp = (C.c_void_p * n)(*[i._h for i in ig]), (C.c_void_p * n)(*[q._h for q in pb])
        cn = (C.c_int * n)(*([B] * n))
        bufs = [[torch.empty_like(t) for t in (x0, u, x, J, lam, d)] for _ in range(n)]
        for bb in bufs: bb[0].fill_(1.0); bb[1].fill_(0.2)
        arr = lambda j: (C.c_void_p * n)(*[bb[j].data_ptr() for bb in bufs])
        A = [arr(j) for j in range(6)]
        torch.cuda.synchronize()
        def mp():
            lib.ocs_multi_compute_states_dev(md2._h, gg, pp, cn, A[0], A[1], A[2], A[3], 0)
            lib.ocs_multi_compute_adjoints_dev(md2._h, gg, pp, cn, A[1], None, A[4], A[5])
        for _ in range(50): mp()
        md2.synchronize(); t0 = time.perf_counter()
        for _ in range(K): mp()
        dt = (time.perf_counter() - t0) / K * 1e6; md2.synchronize()
        print(f"{n} slots on one GPU (persistent worker threads): {dt:.1f} us per pair of calls", flush=True)
