import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdma.so"))
src = torch.arange(512, dtype=torch.float64, device='cuda')
out = torch.zeros(256, dtype=torch.float64, device='cuda')
lib.probe_dma(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), None)
torch.cuda.synchronize()
o = out.cpu().numpy()
print(o[:8], o[120:136], o[184:200])
import numpy as np
exp = np.full(256, -1.0)
for l in range(64):
    exp[2*l] = 2*(63-l); exp[2*l+1] = 2*(63-l)+1
for l in range(32):
    exp[128+2*l] = 128+2*l; exp[128+2*l+1] = 128+2*l+1
print("match", np.array_equal(o, exp))
