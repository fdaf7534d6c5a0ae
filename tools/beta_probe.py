"""Would the residual add belong in the projection GEMM (hipBLASLt beta = 1 with the residual as C) instead of the
LayerNorm kernel that follows? Times mm, addmm(bias), addmm(residual) and a separate add pass at GPT-2 Large shapes.
Measured on MI355X: beta = 1 costs +28 us (k = 1280) / +40-60 us (k = 5120) per GEMM, the fused add in the LayerNorm
kernel ~28 us - not adopted."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from tools.gemm_probe import timeit
M=32768
for (n,k) in [(1280,1280),(1280,5120)]:
    x=torch.randn(M,k,device='cuda',dtype=torch.bfloat16)
    w=torch.randn(n,k,device='cuda',dtype=torch.bfloat16)*0.02
    b=torch.randn(n,device='cuda',dtype=torch.bfloat16)
    res=torch.randn(M,n,device='cuda',dtype=torch.bfloat16)
    out=torch.empty(M,n,device='cuda',dtype=torch.bfloat16)
    wt=w.t()
    t0=timeit(lambda: torch.mm(x,wt,out=out))
    t1=timeit(lambda: torch.addmm(b,x,wt,out=out))
    t2=timeit(lambda: torch.addmm(res,x,wt,out=out))
    t3=timeit(lambda: torch.add(out,res,out=out))
    print(f"n={n} k={k}: mm {t0*1e6:.1f} us | addmm(bias) {t1*1e6:.1f} | addmm(residual, beta=1) {t2*1e6:.1f} | separate add pass {t3*1e6:.1f}")
