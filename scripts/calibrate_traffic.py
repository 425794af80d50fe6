"""Known-byte-count kernels in this path's access pattern (8 B per lane, 512 B per wave
instruction, batch-minor rows) for calibrating FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, section HBM: widths other than 16 B/lane are uncalibrated).
ocs_copy_dev reads and writes every element once with 8-byte accesses: per launch it moves exactly
n*8 bytes in and the same out."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import __graft_entry__ as g

ocs = g.load_package()
lib = sys.modules["ocs_amd._lib"].lib
n = 5 * 1001 * 65536                   # 2.6 GB in + 2.6 GB out: far beyond the 256 MiB Infinity Cache
src = torch.rand(n, dtype=torch.float64, device="cuda")
dst = torch.empty_like(src)
for _ in range(3):
    lib.ocs_copy_dev(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_long(n), None)
torch.cuda.synchronize()
assert torch.equal(src, dst)
print("calibration bytes per launch (read = write):", n * 8)
