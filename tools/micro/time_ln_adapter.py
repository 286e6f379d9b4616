import sys, os
sys.path[:0]=['/root/repo','/root/repo/iccv2025-upp_amd']
import torch
from bench import time_kernel
from upp_hip import ops
import upp_hip.functional as HF
B,Lin,P,D,H=32,85,10,384,32
dev='cuda'
x=torch.randn(B,Lin,D,device=dev); y=torch.randn(B,Lin,D,device=dev); yb=torch.randn(D,device=dev)
g=torch.ones(D,device=dev); bt=torch.zeros(D,device=dev)
W1=torch.randn(H,D,device=dev)*0.05; b1=torch.zeros(H,device=dev); W2=torch.randn(D,H,device=dev)*0.1; b2=torch.zeros(D,device=dev)
Lout=Lin-P
t1=time_kernel(lambda: ops.ln_adapter_fwd(x,y,yb,None,1.0,3,P,g,bt,1e-5,W1,b1,W2,b2,None,0.0,0.7,Lout))
def two():
    xo,h,mean,rstd=ops.rowln_fwd(x,None,None,3,P,y,None,1.0,g,bt,1e-5,Lout,ybias=yb)
    ops.adapter_fwd(h.view(-1,D),xo.view(-1,D),W1,b1,W2,b2,None,0.0,0.7)
t2=time_kernel(two)
print("fused %.2f us   rowln+adapter %.2f us"%(t1*1e3,t2*1e3))

go=torch.randn(B,Lout,D,device=dev)
out,xo,mean,rstd,s1=ops.ln_adapter_fwd(x,y,yb,None,1.0,3,P,g,bt,1e-5,W1,b1,W2,b2,None,0.0,0.7,Lout)
t3=time_kernel(lambda: ops.ln_adapter_bwd_fused(go,xo,mean,rstd,g,bt,s1,W1,W2,None,0.0,0.7,None,1.0,3,P,Lin,True,True,True,True))
def two_b():
    g_ha,part=ops.ln_adapter_bwd(go,xo,mean,rstd,g,bt,s1,W1,W2,None,0.0,0.7)
    ops.rowln_bwd(go,g_ha,xo,mean,rstd,g,3,None,1.0,B,Lin,Lout,D,P,need_x=True,need_prompt=False,need_y=True,need_ln_part=True)
t4=time_kernel(two_b)
print("backward: fused %.2f us   adapter_bwd+rowln_bwd %.2f us"%(t3*1e3,t4*1e3))

# (round 5 timed a tail launch that also computed the next block's head here: 14.14 us against 9.75 + 4.55 in two launches; the entry point
# left the library with ABI 5 -- profiles/r05_time_attention.txt keeps the numbers)
