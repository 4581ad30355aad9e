import os, sys, torch, time
sys.path.insert(0, "/root/repo/multimodal-dynamics_amd")
from mmdyn_hip import ops
sh=(1, 4, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1)
mode,G,Bg,Hi,Wi,Cin,Ho,Wo,N=sh[:9]; Bt=G*Bg
fl=2.0*G*Bg*Ho*Wo*N*16*Cin
s=[torch.cuda.Stream(),torch.cuda.Stream()]
bufs=[(torch.randn(Bt*Hi*Wi*Cin,device="cuda"),torch.randn(16,N,Cin,device="cuda")*0.1,torch.empty(Bt*Ho*Wo,N,device="cuda")) for _ in range(2)]
for reps in (30, 300, 3000, 3000):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps):
        for k,st in enumerate(s):
            with torch.cuda.stream(st):
                ops.B.igemm_nt(bufs[k][0],bufs[k][1],None,bufs[k][2],None,None,None,*sh)
    torch.cuda.synchronize(); t=time.perf_counter()-t0
    print(f"reps {reps}: {t*1e3:8.1f} ms  {2*fl*reps/t/1e12:6.1f} TF/s aggregate", flush=True)
