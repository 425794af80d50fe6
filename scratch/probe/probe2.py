import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
s=torch.cuda.current_stream().cuda_stream
for threads in (16,32,48,64):
  for ilp in (1,4):
    out=torch.zeros(64,dtype=torch.float64,device='cuda'); cyc=torch.zeros(1,dtype=torch.int64,device='cuda')
    iters=1000
    lib.probe_chain(ctypes.c_void_p(out.data_ptr()),ctypes.c_void_p(cyc.data_ptr()),ilp,iters,1,threads,ctypes.c_void_p(s))
    torch.cuda.synchronize()
    c=cyc.item()
    print(f"active lanes={threads} ilp={ilp}: {c/(iters*16*ilp):.2f} ticks/fma/wave")
