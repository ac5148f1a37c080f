"""Diagnostic: what a plain streaming kernel reaches on this box for the traffic mixes of the HBM-bound launches --
read-only, 2 reads : 1 write (the conv1-type input gradient's epilogue streams, the block merge), 1 : 1 -- warm (operands
resident in the 256 MB Infinity Cache from the previous repetition where they fit) and cold (cache flushed by a 1 GiB fill)."""
import sys, os
import torch, numpy as np
DEV = "cuda:0"
flush = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
def t(fn, cold):
    ts = []
    for _ in range(7):
        if cold: flush.fill_(1)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[2:]))
for mb in (26, 51, 103, 206, 411):
    n = mb * 1000 * 1000 // 2
    a = torch.randn(n, device=DEV).bfloat16(); b = torch.randn(n, device=DEV).bfloat16(); c = torch.empty_like(a)
    for name, fn, nbytes in (("sum (read only)", lambda: a.sum(), 2 * n), ("copy (1r:1w)", lambda: c.copy_(a), 4 * n),
                             ("add (2r:1w)", lambda: torch.add(a, b, out=c), 6 * n)):
        w, cd = t(fn, False), t(fn, True)
        print(f"{mb:4d} MB tensors  {name:16s} warm {w:7.1f} us {nbytes / w / 1e6:5.2f} TB/s | cold {cd:7.1f} us {nbytes / cd / 1e6:5.2f} TB/s", flush=True)
