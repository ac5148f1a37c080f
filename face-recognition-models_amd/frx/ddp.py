"""The training step as the product runs it: hipGraph replays on one GPU, and the same stages with
gradient all-reduces between them on several (one process per GPU, torch.distributed backend "nccl" =
RCCL over xGMI).  `DataParallelStep` is the ONE step driver: bench.py, utils.model_utils.train_model and
the multi-process tests all go through it.

The reference has no multi-GPU path at all; semantics chosen here:
  * replicas hold identical weights (broadcast from rank 0 at attach time); each rank owns a shard of the
    global batch;
  * gradients: SUM all-reduce of the flat fp32 gradient buffer in three phases -- the "head" ranges (margin head + fc
    layer: 26 of the 120 MB at 10 575 classes) as soon as the head's and the fc layer's backward are done, i.e. under
    the backward of layer4; the "upper" ranges (layer4, layer3: 72 %) under the backward of layer2 / layer1 / stem;
    then the "lower" rest; few large messages, because xGMI is point-to-point and a ring collective is bound by one
    link; the fused SGD applies grad_scale = 1/world; optionally the buckets travel as bf16 (half the bytes on the
    wire, fp32 accumulation in the update).  Engines without stage_head / stage_upper_rest (and class-sharded heads,
    whose head gradient never travels) keep two phases: upper | lower;
  * BatchNorm statistics, AdaFace's norm EMA and VPL's class memory stay per replica (plain-DDP semantics);
  * CurricularFace's EMA uses the GLOBAL mean target cosine (criterion.py:570-573 on the global batch): one
    1-float all-reduce between the two head phases, which is why the forward stage ends at the cosines.

A step is four stages of the engine (engine.FaceEngine.stage_*): forward | upper | lower | update (data parallel: upper
= head | upper_rest).  Consecutive stages with no exchange between them are captured into ONE hipGraph: on one GPU the
whole step is a single graph; data parallel it is (forward+head) -> (upper) -> (lower) -> (update), or five graphs for
CurricularFace.

The engine is duck-typed (tests drive a small CPU model through the very same class over gloo):
  N, device, world (rw), exchange_ty, ty_sum [1], flat_grads, grad_ranges() -> {"upper": [(lo, hi)..], "lower": [..]},
  stage_forward(images, labels), stage_upper(labels) -> dict, stage_lower(), stage_update(),
  set_lr(lr), pre_step(), post_replay(), replica_state() -> [tensors], after_broadcast(), [graph_key()]
"""
from __future__ import annotations

import torch
import torch.distributed as dist

STAGES = ("forward", "upper", "lower", "update")


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(engine, src=0, group=None):
    """Make every replica start from rank `src`'s parameters, momentum, BN buffers and head state."""
    if _world(group) == 1:
        return
    for t in engine.replica_state():
        dist.broadcast(t, src, group=group)
    engine.after_broadcast()


class DataParallelStep:
    """step(images, labels, lr) -> dict(loss[1], topk[2], ...) of device tensors (valid until the next step)."""

    def __init__(self, engine, group=None, use_graph=True, bf16_buckets=False, broadcast=True, static_inputs=None,
                 split=False):
        """static_inputs: optional (images, labels) tensors the caller fills in place before each step(None, None, lr)
        (bench.py keeps its synthetic batch resident); otherwise step() copies the batch into its own static buffers.
        split: run the multi-GPU structure (segments + collectives) even in a one-rank group (rehearsal on one GPU)."""
        self.eng, self.group = engine, group
        self.world = _world(group)
        self.multi = self.world > 1 or (bool(split) and dist.is_initialized())
        engine.world = self.world
        self.on_gpu = torch.device(engine.device).type == "cuda"
        self.use_graph = bool(use_graph) and self.on_gpu
        self.bf16 = bool(bf16_buckets) and self.multi
        # engines that can finish the head's and the fc layer's gradients early (stage_head / stage_upper_rest) get a bucket
        # of their own for them (class-sharded heads have no head gradient on the wire: they keep the two-bucket plan)
        self.head_bucket = (self.multi and hasattr(engine, "stage_head") and hasattr(engine, "stage_upper_rest")
                            and getattr(engine, "shard", None) is None)
        if hasattr(engine, "set_head_bucket"):       # (sticky engine state: a single-GPU stepper on the same engine resets it;
            engine.set_head_bucket(self.head_bucket)  # both flags are part of the engine's graph_key())
        net = getattr(engine, "net", None)
        if net is not None and hasattr(net, "join_after_upper"):
            net.join_after_upper = self.multi
        self.images = self.labels = None
        if static_inputs is not None:
            self.images, self.labels = static_inputs
        self._graphs = None            # one hipGraph per segment
        self._key = None               # engine.graph_key() the graphs were captured under
        self._warm = False
        self._out = None
        self._pending = []
        self.comm_enabled = True       # False: the segments run, their collectives are skipped (bench.py: exposed-time measurement)
        if self.bf16:
            self._pack = {k: [torch.empty(hi - lo, dtype=torch.bfloat16, device=engine.device) for lo, hi in v]
                          for k, v in engine.grad_ranges().items()}
        if broadcast:
            broadcast_parameters(engine, 0, group)

    # ------------------------------------------------------------------ plan
    def plan(self):
        """[(stage name, compute callable, collective callable or None)]: the compute callables only enqueue device work
        (capturable); a collective is issued by the host after its stage, so it ends a graph segment"""
        if getattr(self.eng, "shard", None) is not None:
            return sharded_plan(self)
        e, multi = self.eng, self.multi

        def upper():
            self._out = e.stage_upper(self.labels)

        def head():
            self._out = e.stage_head(self.labels)
        fwd = ("forward", lambda: e.stage_forward(self.images, self.labels), self._comm_ty if (multi and e.exchange_ty) else None)
        if multi and self.head_bucket:
            # three gradient messages instead of two: head + fc (final right after the head's backward) | layer3-4 | the rest
            return [fwd, ("head", head, self._comm_head), ("upper", e.stage_upper_rest, self._comm_upper),
                    ("lower", e.stage_lower, self._comm_lower), ("update", e.stage_update, None)]
        return [fwd, ("upper", upper, self._comm_upper if multi else None),
                ("lower", e.stage_lower, self._comm_lower if multi else None),
                ("update", e.stage_update, None)]

    def _segment_items(self):
        segs, cur = [], []
        for item in self.plan():
            cur.append(item)
            if item[2] is not None:
                segs.append(cur)
                cur = []
        if cur:
            segs.append(cur)
        return segs

    def segments(self):
        """stage names grouped into graph segments: a segment ends where the host has to issue a collective"""
        return [[name for name, _, _ in items] for items in self._segment_items()]

    def _comm_ty(self):                     # CurricularFace: global sum of the target cosines (criterion.py:570-573)
        if not self.comm_enabled:
            return
        dist.all_reduce(self.eng.ty_sum, op=dist.ReduceOp.SUM, group=self.group)

    def _comm_head(self):                   # head + fc gradients: on the wire while layer4 / layer3 run their backward
        self._pending += self._reduce_ranges("head")

    def _comm_upper(self):                  # starts here, runs under the lower backward
        self._pending += self._reduce_ranges("upper")

    def _comm_lower(self):
        self._pending += self._reduce_ranges("lower")
        self._finish(self._pending)
        self._pending = []

    # ------------------------------------------------------------------ collectives (host-issued, between segments)
    def _reduce_ranges(self, which):
        """start the SUM all-reduce of one set of gradient ranges; returns the handles to finish()"""
        flat = self.eng.flat_grads
        handles = []
        if not self.comm_enabled:
            return handles
        for i, (lo, hi) in enumerate(self.eng.grad_ranges()[which]):
            if self.bf16:
                buf = self._pack[which][i]
                self._cast(flat[lo:hi], buf)
                handles.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, flat[lo:hi]))
            else:
                handles.append((dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))
        return handles

    def _finish(self, handles):
        for work, buf, dst in handles:
            work.wait()                 # NCCL: the current stream waits for the collective (no host block)
            if buf is not None:
                self._cast(buf, dst)

    def _cast(self, src, dst):
        if self.on_gpu:
            from . import ops
            ops.cast(ops.BF16, src, dst, to_f32=dst.dtype == torch.float32)
        else:
            dst.copy_(src)

    # ------------------------------------------------------------------ capture
    def _engine_key(self):
        k = getattr(self.eng, "graph_key", None)
        return k() if k is not None else None

    def _capture(self):
        graphs = []
        torch.cuda.synchronize(self.eng.device)
        for items in self._segment_items():
            g = torch.cuda.CUDAGraph()
            # (thread-local capture mode: RCCL's watchdog thread polls events while this thread captures)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _, fn, _ in items:
                    fn()
            graphs.append(g)
        self._graphs = graphs
        self._key = self._engine_key()

    # ------------------------------------------------------------------ the step
    def step(self, images=None, labels=None, lr=None):
        e = self.eng
        if images is not None:
            if self.images is None:
                dev = self.eng.device
                self.images = torch.empty(images.shape, dtype=images.dtype, device=dev)
                self.labels = torch.empty(labels.shape, dtype=labels.dtype, device=dev)
            if images.dtype != self.images.dtype or images.shape != self.images.shape:
                raise ValueError(f"step planned for {tuple(self.images.shape)} {self.images.dtype} batches, "
                                 f"got {tuple(images.shape)} {images.dtype}")
            self.images.copy_(images, non_blocking=True)
            self.labels.copy_(labels, non_blocking=True)
        if lr is not None:
            e.set_lr(lr)
        e.pre_step()
        # the first step runs eagerly (it is a real step, and it loads every code object); the graphs are captured
        # -- capturing executes nothing -- right before the second one
        if self._graphs is not None and self._key != self._engine_key():
            self._graphs = None        # host-side values baked into the captured launches changed: capture again
        if self.use_graph and self._graphs is None and self._warm:
            self._capture()
        self._pending = []
        for idx, items in enumerate(self._segment_items()):
            if self._graphs is not None:
                self._graphs[idx].replay()
            else:
                for _, fn, _ in items:
                    fn()
            comm = items[-1][2]
            if comm is not None:
                comm()
        if self._graphs is not None:
            e.post_replay()
        self._warm = True
        return self._out

    @property
    def graphed(self):
        return self._graphs is not None


def _reduce_scatter_rows(out, full, group):
    """out [N, D] <- this rank's rows of the SUM over ranks of full [world * N, D]"""
    try:
        dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=group)
    except (RuntimeError, NotImplementedError):          # backends without reduce-scatter (gloo): all-reduce, keep own rows
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        r, n = dist.get_rank(group), out.shape[0]
        out.copy_(full[r * n:(r + 1) * n])


def sharded_plan(st):
    """Step plan for an engine with a CLASS-SHARDED head (engine.shard = (rank, world); SURVEY 8(f)-4, the reference's
    dormant device_id chunking criterion.py:268-278 with the exchanges it lacks).  Each rank owns C / world class columns:
      backbone forward (own batch)            [all-gather features + labels: every rank sees the global batch]
      cosines against the own columns         [all-reduce SUM of the target cosines: the owner of a label fills it in]
      margin + row sweep over the own columns [all-reduce MAX of the row maxima]
      rescale the partial sum-exp             [all-reduce SUM of (sum-exp, rank)]
      loss / top-k, head backward             [reduce-scatter SUM of the partial dL/dfeats: each rank gets its own rows]
      backbone backward / update exactly as in the replicated-head plan; the head gradient never crosses the wire.
    The engine supplies the compute stages (shard_stage_*) and the exchange buffers (feats_l/_g, labels_l/_g, ty_g, part,
    gmax, dx_g, dfeat)."""
    e, g, multi = st.eng, st.group, st.world > 1

    def gather():
        if multi:
            dist.all_gather_into_tensor(e.feats_g, e.feats_l, group=g)
            dist.all_gather_into_tensor(e.labels_g, e.labels_l, group=g)
        else:
            e.feats_g.copy_(e.feats_l)
            e.labels_g.copy_(e.labels_l)

    def ty():
        if multi:
            dist.all_reduce(e.ty_g, op=dist.ReduceOp.SUM, group=g)

    def rmax():
        if multi:
            dist.all_reduce(e.gmax, op=dist.ReduceOp.MAX, group=g)

    def rsum():
        if multi:
            dist.all_reduce(e.part[1:3], op=dist.ReduceOp.SUM, group=g)

    def scatter():
        if multi:
            _reduce_scatter_rows(e.dfeat, e.dx_g, g)
        else:
            e.dfeat.copy_(e.dx_g)

    def head_bwd():
        st._out = e.shard_stage_head_bwd()
    return [("backbone", lambda: e.shard_stage_backbone(st.images, st.labels), gather),
            ("head_cos", e.shard_stage_cos, ty),
            ("head_rows", e.shard_stage_rows, rmax),
            ("head_rescale", e.shard_stage_rescale, rsum),
            ("head_bwd", head_bwd, scatter),
            ("upper", e.shard_stage_upper, st._comm_upper if st.multi else None),
            ("lower", e.stage_lower, st._comm_lower if st.multi else None),
            ("update", e.stage_update, None)]
