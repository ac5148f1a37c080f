"""What does the igemm structure reach on a big, square-ish GEMM (1x1 conv)?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
DEV = "cuda:0"
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, Hi, Ci, Co) in ((256, 8, 2048, 2048), (256, 16, 1024, 1024), (256, 8, 512, 512)):
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); w = (torch.randn(Co, 1, 1, Ci, device=DEV) * 0.05).bfloat16()
    y = torch.empty(N, Hi, Hi, Co, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn_like(y); dw = torch.zeros(Co, 1, 1, Ci, device=DEV)
    t = bench(lambda: ops.conv_fwd(d, x, w, y)); tw = bench(lambda: ops.conv_wgrad(d, x, dy, dw))
    ref = bench(lambda: torch.matmul(x.view(-1, Ci), w.view(Co, Ci).t()))
    fl = ops.conv_flops(d)
    print(f"M={N*Hi*Hi} K={Ci} N={Co}: igemm {t:7.1f} us = {fl/t/1e6:6.1f} TF | wgrad {tw:7.1f} us = {fl/tw/1e6:6.1f} TF | torch.matmul(hipBLASLt) {ref:7.1f} us = {fl/ref/1e6:6.1f} TF")
