"""What a tuned library GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16, fp32 accumulate) reaches on the GEMM shapes the
ResNet-50 convolutions reduce to at batch 256 -- a reference point for k_igemm, which additionally gathers 3x3 taps,
applies the BN prologue and reduces statistics.  M = pixels, N = output channels, K = taps x input channels."""
import torch
DEV = "cuda:0"
SHAPES = [("layer1 3x3", 200704, 64, 576), ("layer2 3x3", 50176, 128, 1152), ("layer3 3x3", 12544, 256, 2304), ("layer4 3x3", 4096, 512, 4608),
          ("layer3 conv1", 12544, 256, 1024), ("layer3 conv3", 12544, 1024, 256), ("layer4 conv1", 4096, 512, 2048), ("layer4 conv3", 4096, 2048, 512),
          ("layer1 conv3", 200704, 256, 64), ("layer2 conv1", 50176, 128, 512)]
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=DEV).bfloat16(); b = torch.randn(N, K, device=DEV).bfloat16(); c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    us = min(t(lambda: torch.matmul(a, b.t(), out=c)) for _ in range(3))
    print(f"{name:14s} M={M:6d} N={N:4d} K={K:4d}  {us:7.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TF/s", flush=True)
