"""CPU, world_size 2 over gloo: the SHIPPED step driver (frx/ddp.py: DataParallelStep -- the class bench.py and
utils.model_utils.train_model run) driving a small CPU model that implements the engine's stage protocol.
What is pinned: the order of stages and collectives, the broadcast of rank 0's state, the two-phase gradient
all-reduce over grad_ranges() with grad_scale 1/world, the CurricularFace-style exchange of the target-cosine sum
between the two head phases, and the bf16 bucket option -- by requiring that two ranks on half batches end up with the
parameters of ONE process stepping the concatenated batch through the same class.
(The GPU engine itself is compared with the single-graph step in tests/test_gpu_stepper.py.)"""
import os

import pytest
import torch
import torch.nn.functional as F


class ToyEngine:
    """x [N,D] -> h = x W1^T -> cos(h, W2 rows) -> logits = 8 cos (1 + t), t an EMA of the GLOBAL mean target cosine
    (stand-in for criterion.py:570-573) -> mean CE.  Flat parameter / gradient / momentum buffers like the engine's:
    [W1 | W2]; W2 is the "upper" range (its gradient is final after stage_upper), W1 the "lower" one."""
    D, H, C = 6, 5, 4

    def __init__(self, N, seed, exchange=True):
        self.N, self.device, self.world, self.exchange_ty = N, torch.device("cpu"), 1, exchange
        g = torch.Generator().manual_seed(seed)
        n1, n2 = self.H * self.D, self.C * self.H
        self.cut = n1
        self.params = torch.randn(n1 + n2, generator=g, dtype=torch.float64) * 0.5
        self.flat_grads = torch.zeros_like(self.params)
        self.mom = torch.zeros_like(self.params)
        self.t = torch.zeros(1, dtype=torch.float64)
        self.ty_sum = torch.zeros(1, dtype=torch.float64)
        self.lr = 0.0
        self.calls = []

    def grad_ranges(self):
        return {"upper": [(self.cut, self.params.numel())], "lower": [(0, self.cut)]}

    def replica_state(self):
        return [self.params, self.mom, self.t]

    def after_broadcast(self):
        self.calls.append("after_broadcast")

    def set_lr(self, lr):
        self.lr = lr

    def pre_step(self):
        self.calls.append("pre")

    def post_replay(self):
        raise AssertionError("no graphs on the CPU")

    def stage_forward(self, x, y):
        self.calls.append("forward")
        self.flat_grads.zero_()
        self.x = x
        W1 = self.params[:self.cut].view(self.H, self.D)
        self.W2 = self.params[self.cut:].view(self.C, self.H).clone().requires_grad_(True)
        self.h = (x @ W1.T).requires_grad_(True)
        self.cos = F.normalize(self.h, dim=1) @ F.normalize(self.W2, dim=1).T
        self.ty_sum[0] = self.cos.detach()[torch.arange(self.N), y].sum()

    def stage_upper(self, y):
        self.calls.append("upper")
        if self.exchange_ty:
            self.t.mul_(0.9).add_(0.1 * self.ty_sum / (self.N * self.world))
        logits = 8.0 * self.cos * (1.0 + self.t)
        loss = F.cross_entropy(logits, y)
        gW2, self.dh = torch.autograd.grad(loss, [self.W2, self.h])
        self.flat_grads[self.cut:] = gW2.reshape(-1)
        return {"loss": loss.detach().reshape(1)}

    def stage_lower(self):
        self.calls.append("lower")
        self.flat_grads[:self.cut] = (self.dh.T @ self.x).reshape(-1)

    def stage_update(self):
        self.calls.append("update")
        g = self.flat_grads / self.world + 5e-4 * self.params
        self.mom.mul_(0.9).add_(g)
        self.params.sub_(self.lr * self.mom)


class ToyEngine3(ToyEngine):
    """the same model for the three-bucket plan: W2's gradient is written in two halves -- the second one by stage_head
    (the "head" ranges: final first, on the wire first), the first one by stage_upper_rest"""

    def _mid(self):
        return self.cut + (self.params.numel() - self.cut) // 2

    def grad_ranges(self):
        return {"head": [(self._mid(), self.params.numel())], "upper": [(self.cut, self._mid())], "lower": [(0, self.cut)]}

    def stage_head(self, y):
        self.calls.append("head")
        if self.exchange_ty:
            self.t.mul_(0.9).add_(0.1 * self.ty_sum / (self.N * self.world))
        logits = 8.0 * self.cos * (1.0 + self.t)
        loss = F.cross_entropy(logits, y)
        gW2, self.dh = torch.autograd.grad(loss, [self.W2, self.h])
        self._gW2 = gW2.reshape(-1)
        m = self._mid() - self.cut
        self.flat_grads[self._mid():] = self._gW2[m:]
        return {"loss": loss.detach().reshape(1)}

    def stage_upper_rest(self):
        self.calls.append("upper_rest")
        m = self._mid() - self.cut
        self.flat_grads[self.cut:self._mid()] = self._gW2[:m]

    def stage_upper(self, y):
        out = self.stage_head(y)
        self.stage_upper_rest()
        return out


def _data(n, steps, seed=7):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(n, ToyEngine.D, generator=g, dtype=torch.float64), torch.randint(0, ToyEngine.C, (n,), generator=g))
            for _ in range(steps)]


def _single(n_total, steps, exchange):
    from frx import ddp
    eng = ToyEngine(n_total, seed=0, exchange=exchange)
    st = ddp.DataParallelStep(eng)
    assert st.segments() == [["forward", "upper", "lower", "update"]] and not st.multi
    losses = [st.step(x, y, 0.05)["loss"].item() for x, y in _data(n_total, steps)]
    return eng, losses


def _worker(rank, world, port, q, exchange, bf16, three=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frx import ddp
    n = 8
    eng = (ToyEngine3 if three else ToyEngine)(n, seed=rank, exchange=exchange)          # ranks start DIFFERENT: the broadcast must fix that
    st = ddp.DataParallelStep(eng, bf16_buckets=bf16)
    assert eng.world == world and eng.calls == ["after_broadcast"]
    if three:       # head + fc bucket of its own: one more exchange point, one more segment
        assert st.head_bucket
        want = [["forward"], ["head"], ["upper"], ["lower"], ["update"]] if exchange else [["forward", "head"], ["upper"], ["lower"], ["update"]]
    else:
        want = [["forward"], ["upper"], ["lower"], ["update"]] if exchange else [["forward", "upper"], ["lower"], ["update"]]
    assert st.segments() == want and st.multi and not st.graphed
    losses = []
    for x, y in _data(n * world, 3):
        sl = slice(rank * n, (rank + 1) * n)
        losses.append(st.step(x[sl], y[sl], 0.05)["loss"].item())
    assert eng.calls[1:7 if three else 6] == (["pre", "forward", "head", "upper_rest", "lower", "update"] if three else ["pre", "forward", "upper", "lower", "update"])
    # numpy arrays travel by value: a torch tensor in an mp queue is a shared-memory handle that dies with this process,
    # which under load can happen before the parent has mapped it
    q.put((rank, eng.params.numpy().copy(), eng.t.numpy().copy(), losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,bf16,three", [(True, False, False), (False, False, False), (False, True, False), (True, False, True),
                                                 (False, True, True)])
def test_data_parallel_step_world2_equals_single_process_on_the_concatenated_batch(exchange, bf16, three):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 3 * exchange + bf16 + 11 * three) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange, bf16, three)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref, ref_losses = _single(16, 3, exchange)
    (_, p0, t0, l0), (_, p1, t1, l1) = [(r, torch.from_numpy(a), torch.from_numpy(b), l) for r, a, b, l in res]
    assert torch.equal(p0, p1) and torch.equal(t0, t1), "replicas diverged"
    tol = 3e-3 if bf16 else 1e-12                    # bf16 buckets round each gradient to 8 bits of mantissa
    assert (p0 - ref.params).abs().max().item() < tol * ref.params.abs().max().item()
    assert (t0 - ref.t).abs().max().item() < 1e-12
    # the global mean loss is the mean of the two shard losses
    for a, b, c in zip(l0, l1, ref_losses):
        assert (a + b) / 2 == pytest.approx(c, rel=5e-3 if bf16 else 1e-10)


def test_segments_follow_the_exchange_points():
    from frx import ddp
    eng = ToyEngine(4, 0, exchange=True)
    st = ddp.DataParallelStep(eng)
    assert st.segments() == [["forward", "upper", "lower", "update"]]      # one GPU: the whole step is ONE graph
    st.multi = True
    assert st.segments() == [["forward"], ["upper"], ["lower"], ["update"]]
    eng.exchange_ty = False
    assert st.segments() == [["forward", "upper"], ["lower"], ["update"]]


# ------------------------------------------------------------------------------------------------------------------
# main_pipeline's data-parallel housekeeping (Weak 11): the min-loss resume prunes the epoch checkpoints
# (model_utils.py:113-117 upstream) -- with several ranks only rank 0 may remove files, and nobody may still be listing
# the directory; wandb.init runs on rank 0 only (main_pipeline), so wandb.log on another rank raises with the real package.
# ------------------------------------------------------------------------------------------------------------------
class _StrictWandb:
    """stand-in for the real package: log() before init() raises (wandb.errors.Error upstream)"""
    def __init__(self):
        self.inited, self.logs = False, []

    def init(self, **kw):
        self.inited = True

    def log(self, d, step=None):
        if not self.inited:
            raise RuntimeError("You must call wandb.init() before wandb.log()")
        self.logs.append(dict(d))


def _housekeeping_worker(rank, world, port, q, ckdir):
    import types
    import torch.distributed as dist
    import torch.nn as nn
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from utils import model_utils as MU
    from utils.schedulers import get_scheduler
    net = nn.Linear(4, 3)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    sch = get_scheduler(opt, "customstep")
    # every rank resumes from the min-loss checkpoint at the same time
    start, loss = MU.load_latest_checkpoint(net, opt, sch, None, ckdir, "M", "cpu", isCheckpoint=False)
    left = sorted(os.listdir(ckdir))
    # the loop's logging: only the rank that called wandb.init may log
    MU.wandb = _StrictWandb()
    if rank == 0:
        MU.wandb.init(project="p")

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = nn.Linear(4, 3)

        def forward(self, x, labels=None):
            z = self.fc(x)
            return [z, z], z.norm(dim=1, keepdim=True), 0, None
    toy = Toy()
    g = torch.Generator().manual_seed(rank)
    batches = [(torch.randn(5, 4, generator=g), torch.randint(0, 3, (5,), generator=g)) for _ in range(3)]
    args = types.SimpleNamespace(lambda_g=0.0, print_freq=1)
    MU._ITERS["n"] = -1
    avg = MU.train_model(toy, batches, nn.CrossEntropyLoss(), torch.optim.SGD(toy.parameters(), lr=0.1), MU.GradScaler(enabled=False),
                         torch.device("cpu"), 1, 1, args)
    q.put((rank, start, loss, left, len(MU.wandb.logs), float(avg)))
    dist.barrier()
    dist.destroy_process_group()


def test_min_loss_resume_and_logging_with_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    import torch.nn as nn
    from utils import model_utils as MU
    from utils.schedulers import get_scheduler
    net = nn.Linear(4, 3)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    sch = get_scheduler(opt, "customstep")
    d = str(tmp_path / "ck")
    for e in range(1, 5):
        MU.save_checkpoint(net, opt, sch, None, 1.0 / e, e, d, "M", isCheckpoint=True)
    MU.save_checkpoint(net, opt, sch, None, 0.25, 4, d, "M", isCheckpoint=False)
    assert len(os.listdir(d)) == 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 11 + 5) % 2000
    procs = [ctx.Process(target=_housekeeping_worker, args=(r, 2, port, q, d)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, start, loss, left, nlogs, avg in res:
        assert (start, loss) == (5, pytest.approx(0.25)), "every rank resumes from the same min-loss checkpoint"
        assert left == ["M_min_loss.pth"], "epoch checkpoints pruned exactly once, after every rank had listed them"
        assert nlogs == (3 if rank == 0 else 0), "only the rank that ran wandb.init logs"
        assert avg == avg
