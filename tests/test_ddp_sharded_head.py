"""CPU, world_size 2 over gloo: the class-sharded head plan (frx/ddp.py: sharded_plan) -- which tensors are
exchanged, with which reduction, in which order -- driven through the shipped DataParallelStep with a small CPU model
that implements the engine's shard stages in closed form.  Two ranks, each owning half of the class columns and half of
the batch, must reproduce ONE process holding every column and stepping the concatenated batch: same loss, same top-1
count, same backbone weights, and the concatenation of their head shards equal to the unsharded head -- within 1e-3
(north star; here 1e-9, float64).  The HIP kernels of the shard phases are checked against the unsharded HIP head in
tests/test_gpu_sharded_head.py."""
import os

import pytest
import torch
import torch.nn.functional as F


class ToyShardEngine:
    """x [N,6] -> feats = x W1^T [N,5] -> CosFace-style head over C classes, s = 8, m = 0.2, columns [c0, c0 + Cl) here."""
    DIN, D, C, S, M = 6, 5, 8, 8.0, 0.2

    def __init__(self, N, rank, world, seed=0):
        self.N, self.device, self.world, self.exchange_ty = N, torch.device("cpu"), 1, False
        self.shard = (rank, world)
        self.Cs = -(-self.C // world)
        self.c0 = rank * self.Cs
        self.Cl = min(self.C, self.c0 + self.Cs) - self.c0
        g = torch.Generator().manual_seed(seed)
        w1 = torch.randn(self.D, self.DIN, generator=g, dtype=torch.float64) * 0.5
        wh = torch.randn(self.C, self.D, generator=g, dtype=torch.float64)          # the SAME full draw on every rank
        self.n1 = w1.numel()
        self.params = torch.cat([w1.reshape(-1), wh[self.c0:self.c0 + self.Cl].reshape(-1),
                                 torch.zeros((self.Cs - self.Cl) * self.D, dtype=torch.float64)])
        self.flat_grads = torch.zeros_like(self.params)
        self.mom = torch.zeros_like(self.params)
        ng = N * world
        f64 = dict(dtype=torch.float64)
        self.labels_l = torch.zeros(N, dtype=torch.int64)
        self.feats_g, self.labels_g = torch.zeros(ng, self.D, **f64), torch.zeros(ng, dtype=torch.int64)
        self.ty_g, self.part, self.gmax = torch.zeros(ng, **f64), torch.zeros(3, ng, **f64), torch.zeros(ng, **f64)
        self.dx_g, self.dfeat = torch.zeros(ng, self.D, **f64), torch.zeros(N, self.D, **f64)
        self.ty_sum = torch.zeros(1, **f64)
        self.lr, self.calls = 0.0, []

    # ---- protocol bits shared with the replicated plan
    def grad_ranges(self):
        return {"upper": [], "lower": [(0, self.n1)]}          # the head columns never travel

    def replica_state(self):
        return [self.params[:self.n1], self.mom[:self.n1]]

    def after_broadcast(self): pass
    def set_lr(self, lr): self.lr = lr
    def pre_step(self): pass
    def post_replay(self): raise AssertionError("no graphs on the CPU")

    def _wh(self):
        return self.params[self.n1:self.n1 + self.Cl * self.D].view(self.Cl, self.D)

    # ---- shard stages (closed form of what the HIP phases compute)
    def shard_stage_backbone(self, x, y):
        self.calls.append("backbone")
        self.flat_grads.zero_()
        self.x = x
        self.feats_l = x @ self.params[:self.n1].view(self.D, self.DIN).T
        self.labels_l.copy_(y)

    def _local(self):
        yl = self.labels_g - self.c0
        owned = (yl >= 0) & (yl < self.Cl)
        return yl, owned

    def shard_stage_cos(self):
        self.calls.append("cos")
        self.f = self.feats_g.clone().requires_grad_(True)
        self.w = self._wh().clone().requires_grad_(True)
        self.cos = F.normalize(self.f, dim=1) @ F.normalize(self.w, dim=1).T
        yl, owned = self._local()
        self.ty_g.zero_()
        rows = torch.nonzero(owned).flatten()
        self.ty_g[rows] = self.cos.detach()[rows, yl[rows]]

    def shard_stage_rows(self):
        self.calls.append("rows")
        yl, owned = self._local()
        z = self.S * self.cos.detach().clone()
        rows = torch.nonzero(owned).flatten()
        z[rows, yl[rows]] -= self.S * self.M
        self.z = z
        self.part[0] = z.max(1).values
        self.part[1] = torch.exp(z - self.part[0][:, None]).sum(1)
        self.part[2] = (self.cos.detach() > self.ty_g[:, None]).sum(1).double()
        self.gmax.copy_(self.part[0])

    def shard_stage_rescale(self):
        self.calls.append("rescale")
        self.part[1] *= torch.exp(self.part[0] - self.gmax)

    def shard_stage_head_bwd(self):
        self.calls.append("head_bwd")
        ng = self.feats_g.shape[0]
        lse = self.gmax + torch.log(self.part[1])
        zy = self.S * (self.ty_g - self.M)
        loss = (lse - zy).mean()
        top1 = int((self.part[2] < 0.5).sum())
        yl, owned = self._local()
        p = torch.exp(self.z - lse[:, None])
        rows = torch.nonzero(owned).flatten()
        p[rows, yl[rows]] -= 1.0
        dcos = p * self.S * (self.world / ng)                    # gout = world: the update rescales by 1 / world
        gf, gw = torch.autograd.grad((dcos * self.cos).sum(), [self.f, self.w])
        self.dx_g.copy_(gf)
        self.flat_grads[self.n1:self.n1 + self.Cl * self.D] = gw.reshape(-1)
        return {"loss": loss.reshape(1), "top1": top1}

    def shard_stage_upper(self):
        self.calls.append("upper")

    def stage_lower(self):
        self.calls.append("lower")
        self.flat_grads[:self.n1] = (self.dfeat.T @ self.x).reshape(-1)

    def stage_update(self):
        self.calls.append("update")
        self.mom.mul_(0.9).add_(self.flat_grads / self.world + 5e-4 * self.params)
        self.params.sub_(self.lr * self.mom)


def _data(n, steps, seed=3):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(n, ToyShardEngine.DIN, generator=g, dtype=torch.float64),
             torch.randint(0, ToyShardEngine.C, (n,), generator=g)) for _ in range(steps)]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frx import ddp
    n = 6
    eng = ToyShardEngine(n, rank, world)
    st = ddp.DataParallelStep(eng)
    assert st.segments() == [["backbone"], ["head_cos"], ["head_rows"], ["head_rescale"], ["head_bwd"], ["upper"], ["lower"], ["update"]]
    outs = []
    for x, y in _data(n * world, 3):
        sl = slice(rank * n, (rank + 1) * n)
        o = st.step(x[sl], y[sl], 0.1)
        outs.append((o["loss"].item(), o["top1"]))
    assert eng.calls[:8] == ["backbone", "cos", "rows", "rescale", "head_bwd", "upper", "lower", "update"]
    # numpy arrays travel by value: a torch tensor in an mp queue is a shared-memory handle that dies with this process,
    # which under load can happen before the parent has mapped it
    q.put((rank, eng.params[:eng.n1].numpy().copy(), eng._wh().numpy().copy(), outs))
    dist.barrier()
    dist.destroy_process_group()


def test_class_sharded_head_world2_equals_the_unsharded_single_process_step():
    import torch.multiprocessing as mp
    from frx import ddp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 11 + 5) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = ToyShardEngine(12, 0, 1)                               # one rank: every column, the whole batch
    st = ddp.DataParallelStep(ref)
    ref_out = []
    for x, y in _data(12, 3):
        o = st.step(x, y, 0.1)
        ref_out.append((o["loss"].item(), o["top1"]))
    (_, w1a, wha, oa), (_, w1b, whb, ob) = [(r, torch.from_numpy(a), torch.from_numpy(b), o) for r, a, b, o in res]
    assert torch.equal(w1a, w1b), "backbone replicas diverged"
    assert (w1a - ref.params[:ref.n1]).abs().max().item() < 1e-9
    assert (torch.cat([wha, whb]) - ref._wh()).abs().max().item() < 1e-9        # shards side by side == the unsharded head
    for (la, ta), (lb, tb), (lr_, tr) in zip(oa, ob, ref_out):
        assert la == pytest.approx(lr_, abs=1e-9) and lb == pytest.approx(lr_, abs=1e-9)      # (north star: 1e-3)
        assert ta == tb == tr
