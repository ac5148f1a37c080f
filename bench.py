#!/usr/bin/env python3
"""Headline benchmark: images/s of one ArcFace ResNet-50 TRAINING step (forward + CE + backward +
SGD) at 112x112, bf16, batch 256 per GPU, 10 575 identities -- BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU (RCCL over xGMI via torch.distributed "nccl").  Rank 0 prints ONE JSON line.
  value        whole-job images/s, inputs resident in HBM, max-over-ranks wall time of K steps
  roofline     the dominant kernel CLASS (by device time; a class = one k_igemm instantiation: tile, gather mode, prologue,
               epilogue, staging -- frx_last_conv_launch) timed per launch with HIP events on its own stream in a separate
               eager pass after the timed region (the timed region replays a hipGraph, which leaves no place for
               per-kernel events); achieved = algorithmic bytes (or FLOPs) / duration; per_class lists every class
  rccl         (N > 1) ranks that joined, each gradient bucket's all-reduce timed alone, and the time the collectives
               leave exposed (steps with collectives - steps without)
  cpu_baseline the CPU oracle (oracle/resnet50.py, "port") on this box's host cores:
               BASELINE.json configs[0] (ArcFace R50, 100 identities, bs 32, fp32), a few steps
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))

import torch  # noqa: E402

FLOP_PER_IMG_BACKBONE = 6.4025e9        # fwd + dgrad + wgrad, no image gradient
PEAK_BF16_TFLOPS = 2516.6               # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU dense (MI355X_MICROARCH.md: ~2.5 PF)
PEAK_HBM_GBS = 8000.0


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))          # a 1-GPU box's CPU share


def cpu_baseline(threads, max_seconds=25.0):
    from oracle import heads as H
    from oracle.resnet50 import FaceNet, make_sgd, train_step
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = FaceNet(H.ARC, 100)
    opt = make_sgd(net, 0.1)
    g = torch.Generator().manual_seed(1234)
    images = torch.rand(32, 3, 112, 112, generator=g) * 2 - 1
    labels = torch.randint(0, 100, (32,), generator=g)
    train_step(net, opt, images, labels)                       # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 2 or time.perf_counter() - t0 < max_seconds / 2:          # a bounded ~12 s sample of CPU work
        train_step(net, opt, images, labels)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(32 * n / dt, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n} timed steps (+1 warm-up) of BASELINE configs[0]: ArcFace ResNet-50, 100 identities, bs=32, "
                      f"fp32, torch CPU oracle, {dt:.1f}s"}


def baseline_config(args, world):
    """which BASELINE.json config (if any) the arguments of this run describe -- the workload string follows the run"""
    key = (args.head, args.classes, args.batch, args.dtype, bool(args.shard_head))
    if key == ("arcface", 10575, 256, "bf16", False):
        return "BASELINE configs[1]" + ("" if world == 1 else f" per-GPU shape on {world} GPUs")
    if key == ("cosface", 10575, 256, "bf16", False):
        return "BASELINE configs[2]" + (" per-GPU shape" if world != 8 else "")
    if args.head == "curricular" and args.classes in (85000, 85742) and args.batch == 128 and args.dtype == "bf16":
        return "BASELINE configs[3]" + (" per-GPU shape" if world != 8 else "") + (", class-sharded head" if args.shard_head else "")
    return "not a BASELINE config (custom --head / --classes / --batch / --dtype)"


def kernel_pass(eng, images, labels, steps=3):
    """Per-launch HIP-event timing of the GEMM-class kernels over `steps` eager steps.  Each launch is bracketed by
    events on its own stream; a launch's duration is the MINIMUM over the steps (an eager step is host-paced, so a
    single sample can include a host hiccup between the first event and the launch).  The totals are scaled
    back to `steps` steps so that per-step figures read naturally."""
    from frx import ops
    # (a launch is timed ALONE: the side stream that runs the projection branches next to the conv chain in the timed step
    # would charge each launch for its neighbour here -- in line for this pass)
    side, eng.net.branch_stream = eng.net.branch_stream, None
    net = eng.net

    def eager_step(check=False):
        # the step's stages, eagerly (the fused update zeroes the gradients it consumes: they are inspected before it)
        eng.pre_step()
        eng.stage_forward(images, labels)
        eng.stage_upper(labels)
        eng.stage_lower()
        if check:
            # work integrity: the step must have produced a weight gradient for EVERY conv layer (a kernel that silently
            # does nothing makes the benchmark faster, not slower -- this is what would show it)
            dead = [c.name for c in net.convs if float(net.w_grad(c).abs().max()) == 0.0]
            if dead:
                raise SystemExit(f"bench: integrity check failed: zero weight gradient in {len(dead)} conv layers, e.g. {dead[:4]}")
        eng.stage_update()
    eager_step()
    torch.cuda.synchronize()
    per_step = []
    for i in range(steps):
        ops.PROFILER = []
        eager_step(check=(i == steps - 1))
        torch.cuda.synchronize()
        per_step.append([(label, flops, e0.elapsed_time(e1) * 1e-3, nbytes) for label, flops, e0, e1, nbytes in ops.PROFILER])
    ops.PROFILER = None
    eng.net.branch_stream = side
    agg = {}
    for calls in zip(*per_step):
        label, flops, _, nbytes = calls[0]
        secs = min(c[2] for c in calls)
        a = agg.setdefault(label, [0.0, 0.0, 0, 0.0])
        a[0] += secs * steps
        a[1] += flops * steps
        a[2] += steps
        a[3] += nbytes * steps
    return agg


def rccl_report(stepper, step, args, ms_with, dev):
    """Self-diagnosis of a multi-GPU run (every rank runs the same sequence; rank 0 reports): the ranks that joined, each
    gradient bucket's all-reduce timed ALONE (barrier, then 5 back-to-back calls between two events), the step without
    its collectives (same graph segments, collectives skipped), and what the collectives leave exposed = the difference."""
    import torch.distributed as dist
    eng, k = stepper.eng, min(10, max(3, args.steps))
    out = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "bf16_buckets": bool(stepper.bf16), "allreduce_ms": {},
           "bucket_mbytes": {}}
    torch.cuda.synchronize()
    for name, ranges in eng.grad_ranges().items():
        bufs = [torch.zeros(hi - lo, device=dev, dtype=torch.bfloat16 if stepper.bf16 else torch.float32) for lo, hi in ranges]
        for b in bufs:
            dist.all_reduce(b)                                  # (connection set-up outside the timing)
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for b in bufs:
                dist.all_reduce(b)
        e1.record()
        torch.cuda.synchronize()
        out["allreduce_ms"][name] = round(e0.elapsed_time(e1) / 5, 3)
        out["bucket_mbytes"][name] = round(sum(b.numel() * b.element_size() for b in bufs) / 1e6, 1)
    stepper.comm_enabled = False                                # same segments, no collective issued (weights drift apart:
    dist.barrier()                                              # measured LAST, nothing trains on them afterwards)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        step(i)
    torch.cuda.synchronize()
    ms_without = (time.perf_counter() - t0) / k * 1e3
    stepper.comm_enabled = True
    tt = torch.tensor([ms_without], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    out["step_ms_without_collectives"] = round(float(tt.item()), 3)
    out["exposed_ms"] = round(ms_with - float(tt.item()), 3)
    return out


def main():
    # Everything except the final JSON line goes to stderr -- including what native libraries (RCCL prints a
    # version banner at init) write to file descriptor 1.
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))      # (rank 0 of the children prints the JSON line)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        result = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if result is not None:
        print(json.dumps(result), flush=True)


def self_launch(args_list, gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as children of a
    torch.distributed.run child process BEFORE this process touches the GPU (it never does), pass the
    child's output through and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args_list
    log("launching " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--classes", type=int, default=10575)
    ap.add_argument("--head", default="arcface",
                    help="arcface | cosface | sphereface | curricular | mv_am | mv_arc | adaface | elastic_arc | elastic_cos | magface | vpl_arcface")
    ap.add_argument("--lambda-g", type=float, default=0.0, help="MagFace: weight of loss_g (model_utils.py:180, 482)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--split", action="store_true",
                    help="run the data-parallel step structure (graph segments + RCCL all-reduces between them) on one GPU")
    ap.add_argument("--bf16-buckets", action="store_true", help="data parallel: gradients travel as bf16")
    ap.add_argument("--shard-head", action="store_true",
                    help="class-sharded head (SURVEY 8f-4): each rank owns classes/N columns, no head gradient on the wire")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.split:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from frx import ddp, engine as E, ops
    dt = ops.BF16 if args.dtype == "bf16" else ops.F32
    if args.shard_head and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    eng = E.FaceEngine(args.head, args.classes, args.batch, dtype=dt, device=dev, seed=0, lambda_g=args.lambda_g,
                       shard=(rank, world) if args.shard_head else None)
    g = torch.Generator().manual_seed(1234 + rank)
    nb = 4
    batches = [((torch.rand(args.batch, 3, 112, 112, generator=g) * 2 - 1).to(dev),
                torch.randint(0, args.classes, (args.batch,), generator=g).to(dev)) for _ in range(nb)]
    images = torch.empty_like(batches[0][0])
    labels = torch.empty_like(batches[0][1])
    # The reference's default lr 0.1 (model_utils.py:480) assumes ImageNet-pretrained weights; from the random init
    # used here (and random labels) it diverges to NaN within ~20 steps, and 0.02 still does after a few hundred:
    # the benchmark trains at 0.005, where the loss falls monotonically over 300 steps.

    def feed(i):
        images.copy_(batches[i % nb][0])
        labels.copy_(batches[i % nb][1])

    # The step driver is the product's own (frx/ddp.py, also behind utils.model_utils.train_model): one hipGraph
    # on one GPU; graph segments with the gradient all-reduces between them when data parallel.  Rank 0's
    # parameters are broadcast at construction.
    stepper = ddp.DataParallelStep(eng, use_graph=not args.no_graph, bf16_buckets=args.bf16_buckets,
                                   static_inputs=(images, labels), split=args.split)
    log(f"engine ready: {eng.net.n_params} parameters, batch {args.batch}, world {world}")

    def step(i):
        feed(i)
        return stepper.step(None, None, args.lr)

    graph = None
    for i in range(max(args.warmup, 2)):           # (step 0 is eager, the graphs are captured before step 1)
        step(i)
    graph = stepper.graphed
    log(f"{len(stepper.segments())} graph segment(s) captured" if graph else "eager mode")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    loss_trace = torch.zeros(args.steps, device=dev)      # one 4-byte device copy per step, read after the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        o = step(i)
        loss_trace[i].copy_(o["loss"].reshape(()))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_s = time.perf_counter() - t0
    loss = float(o["loss"].item())
    log("loss per step: " + " ".join(f"{v:.1f}" for v in loss_trace.tolist()))
    log(f"timed region done: {dt_s / args.steps * 1e3:.3f} ms/step")
    tmax = torch.tensor([dt_s], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_s = float(tmax.item())

    rccl = None
    if world > 1 or args.split:
        try:
            rccl = rccl_report(stepper, step, args, dt_s / args.steps * 1e3, dev)
        except Exception as exc:                # a diagnostic must never cost the run its result line
            rccl = {"ranks": world, "error": repr(exc)}
    result = None
    if rank == 0:
        ips = world * args.batch * args.steps / dt_s
        flop_img = FLOP_PER_IMG_BACKBONE + 6.0 * 512 * args.classes
        agg = kernel_pass(eng, images, labels) if (world == 1 and not args.shard_head) else {}      # (plain single-GPU eager step)
        roof = None
        if agg:
            label, (secs, flops, launches, nbytes) = max(agg.items(), key=lambda kv: kv[1][0])
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
            ach_tf, ach_gbs = flops / secs / 1e12, nbytes / secs / 1e9
            # roofline model: the kernel's algorithmic intensity against the ridge point decides which roof binds
            intensity, ridge = flops / nbytes, peak * 1e12 / (PEAK_HBM_GBS * 1e9)
            # HBM bytes per launch of that kernel from the PMC passes of scripts/collect_traffic.sh (rocprofv3 cannot
            # run inside this process); null until such a summary has been committed under profiles/
            traffic = traffic_source = None
            tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tf):
                with open(tf) as fh:
                    tj = json.load(fh)
                traffic = tj.get(label, {}).get("hbm_bytes_per_launch")
                if traffic is not None:        # where the number comes from: it is NOT measured by this run
                    traffic_source = {"file": "profiles/traffic_latest.json", **tj.get("_meta", {"collected": "round 1 (r01_f)"})}
            common = {"kernel": label, "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": round(secs / launches * 1e6, 2),
                      "launches_per_step": launches // 3, "flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                      "mfma_tflops": round(ach_tf, 2), "mfma_frac": round(ach_tf / peak, 4),
                      "hbm_gbs": round(ach_gbs, 1), "hbm_frac": round(ach_gbs / PEAK_HBM_GBS, 4),
                      "algorithmic_bytes_per_launch": round(nbytes / launches), "algorithmic_flops_per_launch": round(flops / launches),
                      # every GEMM-class behaviour of the step, largest first: where the time is and against which roof
                      "per_class": {k: {"ms_per_step": round(v[0] / 3 * 1e3, 3), "tflops": round(v[1] / v[0] / 1e12, 1),
                                        "tbs": round(v[3] / v[0] / 1e12, 2), "launches_per_step": v[2] // 3,
                                        "bound": "hbm" if v[1] / v[3] < ridge else "mfma",
                                        "frac": round((v[3] / v[0] / 1e9 / PEAK_HBM_GBS) if v[1] / v[3] < ridge else (v[1] / v[0] / 1e12 / peak), 4)}
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}}
            if intensity < ridge:
                roof = {"bound": "hbm", "achieved": round(ach_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach_gbs / PEAK_HBM_GBS, 4), **common}
            else:
                roof = {"bound": "mfma", "achieved": round(ach_tf, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach_tf / peak, 4), **common}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle, configs[0]) ...")
            cpu = cpu_baseline(host_threads())
        result = {
            "metric": "images/sec (112x112) ArcFace ResNet-50 train",
            "value": round(ips, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_s / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.head} ResNet-50 (torchvision v1.5 topology, fc->512), {args.classes}-class head, "
                                   f"bs={args.batch}/GPU, 112x112, fwd+CE+bwd+SGD(momentum 0.9, wd 5e-4), "
                                   f"random-init weights, {baseline_config(args, world)}",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "hip_graph": bool(graph), "graph_segments": len(stepper.segments()) if graph else 0,
                       "bf16_grad_buckets": bool(stepper.bf16), "class_sharded_head": bool(args.shard_head),
                       "lr": args.lr, "final_loss": round(loss, 4)},
            "step_mfma_frac": round(ips * flop_img / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if rccl is not None:
            result["rccl"] = rccl
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return result if rank == 0 else None


if __name__ == "__main__":
    main()
