"""What HBM delivers on this box for plain streams (torch kernels): copy, fill (write only), sum (read only); 1 GiB arrays.
python scripts/probe/hbm_rates.py"""
import torch
dev = torch.device('cuda:0')
n = 1 << 27   # doubles: 1 GiB
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
b = n * 8
for sz in (n, n // 8):   # 1 GiB and 128 MiB (the size class of one pass's arrays; fits the 256 MB memory-side cache)
    xs, ys = x[:sz], y[:sz]
    bb = sz * 8
    print(f"{bb/2**20:.0f} MiB: copy {2*bb/t(lambda: ys.copy_(xs))/1e12:.2f} TB/s (read + write), "
          f"fill {bb/t(lambda: ys.fill_(1.0))/1e12:.2f} TB/s, sum {bb/t(lambda: xs.sum())/1e12:.2f} TB/s", flush=True)
