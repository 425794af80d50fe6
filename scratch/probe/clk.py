import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libclk.so"))
s=torch.cuda.current_stream().cuda_stream
for blocks,threads in ((64,64),(256,64),(1024,64),(256,256),(1024,256),(2048,256)):
  for rep in range(3):
    out=torch.zeros(blocks*threads,dtype=torch.float64,device='cuda'); res=torch.zeros(2*blocks,dtype=torch.int64,device='cuda')
    iters=20000
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); lib.probe_clk(ctypes.c_void_p(out.data_ptr()),ctypes.c_void_p(res.data_ptr()),iters,blocks,threads,ctypes.c_void_p(s)); e1.record()
    torch.cuda.synchronize()
    r=res.view(-1,2).double()
    cyc=r[:,0].median().item(); rt=r[:,1].median().item()
    print(f"blocks={blocks} threads={threads}: {cyc/(iters*64):.2f} cyc/fma, clock {cyc/rt*100:.0f} MHz, kernel {e0.elapsed_time(e1)*1e3:.0f} us")
