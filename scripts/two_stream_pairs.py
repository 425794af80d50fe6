"""Two independent batches (two integrator handles, own outputs) evaluated back to back on ONE stream against the same
work on TWO streams (one batch each): does the chain-bound state pass of one batch run under the HBM-bound adjoint pass of
the other?  python scripts/two_stream_pairs.py   (NS / BATCH in the environment)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
dev = torch.device('cuda:0')
nS, N, batch = int(os.environ.get('NS', '4')), 1000, int(os.environ.get('BATCH', '4096'))
m = [3.0, 2.5, 2.0, 1.5][:nS]
prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
tspan = np.linspace(0, 10, N + 1)
sets = []
for k in range(2):
    integ = ocs.RK4Integrator(tspan)
    x0 = torch.ones((nS, batch), dtype=torch.float64, device=dev)
    u = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, batch), dtype=torch.float64, device=dev)
    x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    sets.append((integ, x0, u, x, torch.empty_like(x), torch.empty_like(u)))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def pair(k):
    integ, x0, u, x, lam, d = sets[k]
    integ.compute_states_dev(prob, x0, u, x); integ.compute_adjoints_dev(prob, u, None, lam, d)
def run(two, K=40):
    for _ in range(3):
        for k in range(2): pair(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        for k in range(2):
            if two:
                with torch.cuda.stream(streams[k]): pair(k)
            else:
                pair(k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (2 * K) * 1e6
for rep in range(2):
    print(f"one stream: {run(False):.1f} us per pass pair   two streams: {run(True):.1f} us per pass pair", flush=True)
