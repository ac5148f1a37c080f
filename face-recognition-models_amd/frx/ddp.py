"""Data-parallel glue: one process per GPU, gradients all-reduced over RCCL (torch.distributed
backend "nccl" on ROCm) across xGMI.  The reference has no multi-GPU path at all;
semantics chosen here:
  * replicas hold identical weights; each rank owns a shard of the global batch;
  * one SUM all-reduce of the flat fp32 gradient buffer per step, issued as a few large buckets
    (xGMI is point-to-point: few large messages), then the fused SGD applies grad_scale = 1/world;
  * BatchNorm statistics stay per replica (plain-DDP semantics);
  * CurricularFace's EMA uses the GLOBAL mean target cosine: one extra 1-float all-reduce.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def bucket_ranges(numel, n_buckets):
    """Split [0, numel) into n_buckets contiguous 256-element-aligned ranges (last first: the head /
    fc / layer4 gradients are complete first in backward order)."""
    n_buckets = max(1, int(n_buckets))
    step = (numel + n_buckets - 1) // n_buckets
    step = (step + 255) // 256 * 256
    out, lo = [], 0
    while lo < numel:
        hi = min(numel, lo + step)
        out.append((lo, hi))
        lo = hi
    return out[::-1]


class GradAllReducer:
    """Callable(flat_grads): in-place SUM all-reduce across the process group.
    `via_host` stages through pinned host memory for the gloo backend (CPU rehearsal of the N>1 path)."""

    def __init__(self, group=None, n_buckets=4, via_host=False):
        self.group, self.n_buckets, self.via_host = group, n_buckets, via_host
        self.world = dist.get_world_size(group)

    def __call__(self, flat):
        if self.world == 1:
            return
        if self.via_host:
            h = flat.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(h)
            return
        works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for lo, hi in bucket_ranges(flat.numel(), self.n_buckets)]
        for w in works:
            w.wait()


class TargetCosineAllReducer:
    """Callable(tensor[1]) -> global sample count; sums CurricularFace's sum(t_y) over ranks."""

    def __init__(self, local_batch, group=None, via_host=False):
        self.group, self.via_host = group, via_host
        self.count = local_batch * dist.get_world_size(group)

    def __call__(self, ty_sum):
        if self.via_host:
            h = ty_sum.detach().cpu()
            dist.all_reduce(h, group=self.group)
            ty_sum.copy_(h)
        else:
            dist.all_reduce(ty_sum, group=self.group)
        return self.count


def attach(engine, group=None, n_buckets=4, via_host=False):
    """Turn a FaceEngine into one data-parallel replica (weights must already be identical)."""
    engine.world = dist.get_world_size(group)
    engine.allreduce = GradAllReducer(group, n_buckets, via_host)
    from . import ops
    if engine.kind == ops.CURR:
        engine.ty_allreduce = TargetCosineAllReducer(engine.N, group, via_host)
    return engine


def broadcast_parameters(engine, src=0, group=None, via_host=False):
    """Make every replica start from rank `src`'s parameters, momentum and BN buffers."""
    for t in (engine.net.params, engine.net.mom, engine.net.running_mean, engine.net.running_var, engine.t):
        if via_host:
            h = t.detach().cpu()
            dist.broadcast(h, src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src, group=group)
    engine.net.sync_weights()
