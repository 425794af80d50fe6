import torch, ctypes, os, time
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
print("torch", torch.__version__, torch.cuda.get_device_name(0))
n=1<<20
x=torch.arange(n,dtype=torch.float64,device='cuda'); y=torch.ones(n,dtype=torch.float64,device='cuda')
s=torch.cuda.current_stream().cuda_stream
r=lib.probe_axpy(ctypes.c_void_p(x.data_ptr()),ctypes.c_void_p(y.data_ptr()),ctypes.c_double(2.0),ctypes.c_int(n),ctypes.c_void_p(s))
torch.cuda.synchronize()
print("axpy rc",r, "ok", bool(torch.equal(y, 2*torch.arange(n,dtype=torch.float64,device='cuda')+1)))
for threads in (64,256,512):
  for ilp in (1,2,4,8):
    out=torch.zeros(threads,dtype=torch.float64,device='cuda'); cyc=torch.zeros(1,dtype=torch.int64,device='cuda')
    iters=1000
    lib.probe_chain(ctypes.c_void_p(out.data_ptr()),ctypes.c_void_p(cyc.data_ptr()),ilp,iters,1,threads,ctypes.c_void_p(s))
    torch.cuda.synchronize()
    c=cyc.item()
    print(f"threads={threads} ilp={ilp}: {c/(iters*16):.2f} ticks per round of {ilp} fma -> {c/(iters*16*ilp):.2f} ticks/fma/wave")
import subprocess
print(subprocess.run("nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c 'import os;print(len(os.sched_getaffinity(0)))'",shell=True,capture_output=True,text=True).stdout)
