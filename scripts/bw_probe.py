"""Probe: achievable HBM write / copy bandwidth on this box for activation-sized tensors."""
import torch, time
DEV = "cuda:0"
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (26, 103, 411):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=DEV); b = torch.empty_like(a)
    t_fill = bench(lambda: a.fill_(1.0)); t_copy = bench(lambda: b.copy_(a)); t_read = bench(lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32))
    print(f"{mb:4d} MB: fill {t_fill:7.1f} us = {mb*1.048576/t_fill*1e3:6.0f} GB/s | copy {t_copy:7.1f} us = {2*mb*1.048576/t_copy*1e3:6.0f} GB/s (r+w) | sum {t_read:7.1f} us = {mb*1.048576/t_read*1e3:6.0f} GB/s")
