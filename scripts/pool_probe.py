import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "face-recognition-models_amd"))
import torch
from frx import ops
N=256; H=56; C=64; DEV="cuda:0"
y=torch.randn(N,H,H,C,device=DEV).bfloat16(); sc=torch.rand(C,device=DEV)+0.5; sh=torch.randn(C,device=DEV)*0.1
out=torch.empty(N,28,28,C,device=DEV,dtype=torch.bfloat16); arg=torch.empty(N,28,28,C,device=DEV,dtype=torch.uint8)
g=torch.randn_like(out); dpost=torch.empty_like(y)
def t(fn,reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/reps
print("pool_fwd %.1f us  pool_bwd %.1f us" % (t(lambda: ops.stem_pool_fwd(ops.BF16,N,H,H,C,y,sc,sh,out,arg)), t(lambda: ops.stem_pool_bwd(ops.BF16,N,H,H,C,g,arg,dpost))))
