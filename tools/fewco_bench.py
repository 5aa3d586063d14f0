import sys, os, math, torch
sys.path.insert(0, os.getcwd())
from babe_amd import ops
B=int(os.environ.get("B","2"))
Ns=[64,96,96,128,128,256,256]
for i in range(7):
    F, T = 64*(i+1), (4096>>i) if i==6 else (4096>>i)//2
    N=Ns[i]
    w=torch.randn(N,2,5,3,device="cuda")/5
    pc=ops.PackedConv(w)
    gy=torch.randn(B,N,F,T,device="cuda"); out=torch.empty(B,2,F,T,device="cuda")
    for mode in ("fewco","mfma"):
        ops.FEWCO = mode=="fewco"
        for _ in range(2): ops.conv2d(gy,pc,out,transpose=True)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d(gy,pc,out,transpose=True)
        e1.record(); torch.cuda.synchronize()
        print(f"level {i} N={N} F={F} T={T} {mode}: {e0.elapsed_time(e1)/10*1e3:.1f} us")
