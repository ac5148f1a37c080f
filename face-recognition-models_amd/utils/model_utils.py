"""Train / evaluate API with the reference's names and signatures
(main_code/utils/model_utils.py:43-138, 147-216, 320-474, 476-590), driving the MI355X-native engine.

What changed relative to upstream, on purpose:
  * throughput is measured (upstream divides by ~0, :196-198);
  * host<->device syncs happen every `print_freq` steps, not 4+ times per step: loss / top-k are
    accumulated on the device;
  * verification embeds every image ONCE and re-uses the cosines for all folds (upstream forwards each
    pair three times per fold: 19x redundant);
  * `compute_auc` works (upstream never imports roc_auc_score, M5);
  * wandb / dotenv are optional.
What is preserved: step ordering (zero_grad -> forward -> CE -> backward -> SGD), loss averaging, the
10-fold protocol exactly as shipped (threshold tuned on the held-out fold, accuracy on the other nine,
strict `>`), checkpoint file names / dict keys / keep-3 rotation, argparse flags.
"""
from __future__ import annotations

import argparse
import os
import re
import shutil
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import ConcatDataset, DataLoader

from frx import ops, verify as V
from frx.module import NativeFaceNet

from .config import *  # noqa: F401,F403  (reference scripts rely on the star import)
from .config import BACKBONE, WORKING_PATH
from .dataset import CASIAwebfaceDataset, FlatPairDataset
from .metrics import accuracy
from .schedulers import get_scheduler
from .utils import AverageMeter, ProgressMeter

try:                                     # optional experiment tracker (absent offline)
    import wandb
except Exception:                        # pragma: no cover
    class _NoWandb:
        def init(self, **k): return None
        def log(self, *a, **k): return None
        def save(self, *a, **k): return None
        def finish(self): return None
    wandb = _NoWandb()

try:
    from torch.amp import GradScaler
except ImportError:                      # pragma: no cover
    from torch.cuda.amp import GradScaler


# ---------------------------------------------------------------------------------------------- optimiser
class FusedSGD(torch.optim.Optimizer):
    """optim.SGD(lr, momentum 0.9, weight_decay 5e-4) (model_utils.py:557 upstream) as ONE kernel launch over
    the engine's flat parameter / gradient / momentum buffers.  Keeps torch's optimizer interface
    (param_groups for the LR scheduler, state_dict for checkpoints)."""

    def __init__(self, model: NativeFaceNet, lr, momentum=0.9, weight_decay=5e-4):
        self.model = model
        # (every key torch.optim.SGD keeps in a param group, so that optimizer_state_dict written here loads into
        # upstream's optim.SGD and steps there: model_utils.py:126-132 of the reference)
        super().__init__(list(model.parameters()), dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay,
                                                        nesterov=False, maximize=False, foreach=None, differentiable=False,
                                                        fused=None))
        self._pending_mom = None

    def _engine(self):
        eng = self.model._primary
        if eng is None:
            raise ops.FrxError("FusedSGD.step() before the model's first forward on the HIP device")
        return eng

    def zero_grad(self, set_to_none=True):
        eng = self.model._primary
        if eng is not None:
            eng.net.zero_grad()
        super().zero_grad(set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        eng = self._engine()
        g = self.param_groups[0]
        self._apply_pending(eng)
        # (torch semantics: .grad survives optimizer.step(); the step driver's own update zeroes it instead of a fill launch)
        eng.net.sgd_step(g["lr"], g["momentum"], g["weight_decay"], zero_grads=False)
        self.model._synced_version = self.model._version_sum()

    # checkpoints: momentum lives in the engine's flat buffer; it is saved and loaded in torch.optim.SGD's OWN format
    # (state[i]["momentum_buffer"] per parameter, in the parameter's torch shape), so an optimizer_state_dict written
    # here resumes upstream's optim.SGD and one written upstream resumes here (model_utils.py:58-65, 126-132)
    def state_dict(self):
        sd = super().state_dict()
        if self.model._primary is not None:
            sd["state"] = {i: {"momentum_buffer": p._frx_mom.detach().clone().contiguous()}
                           for i, p in enumerate(self.param_groups[0]["params"])}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        legacy = sd.pop("frx_momentum", None)               # round-1 checkpoints: the flat buffer as one blob
        state = sd.get("state") or {}
        self._pending_mom = {int(i): st["momentum_buffer"] for i, st in state.items()
                             if isinstance(st, dict) and st.get("momentum_buffer") is not None}
        if not self._pending_mom:
            self._pending_mom = legacy
        sd["state"] = {}
        super().load_state_dict(sd)
        if self.model._primary is not None:
            self._apply_pending(self.model._primary)

    def _apply_pending(self, eng):
        pend, self._pending_mom = self._pending_mom, None
        if pend is None:
            return
        if isinstance(pend, dict):
            params = self.param_groups[0]["params"]
            eng.net.mom.zero_()                             # (parameters without a buffer yet, the stem's padding slots)
            for i, buf in pend.items():
                dst = params[i]._frx_mom
                if tuple(buf.shape) != tuple(dst.shape):
                    raise ValueError(f"optimizer state: momentum_buffer {i} has shape {tuple(buf.shape)}, "
                                     f"the parameter has {tuple(dst.shape)}")
                dst.copy_(buf.to(eng.device, torch.float32))
        else:
            eng.net.mom.copy_(pend.to(eng.device))


# ---------------------------------------------------------------------------------------------- checkpoints
def save_checkpoint(model, optimizer, scheduler, scaler, train_loss, epoch, model_checkpoints_path, model_name,
                    isCheckpoint=True):
    """{name}_checkpoint_epoch_{e}.pth keeping the newest three, or {name}_min_loss.pth (model_utils.py:43-81)."""
    os.makedirs(model_checkpoints_path, exist_ok=True)
    blob = {"epoch": epoch, "train_loss": train_loss, "model_state_dict": model.state_dict(),
            "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else None,
            "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
            "scaler_state_dict": scaler.state_dict() if scaler is not None else None}
    if not isCheckpoint:
        torch.save(blob, os.path.join(model_checkpoints_path, f"{model_name}_min_loss.pth"))
        return
    torch.save(blob, os.path.join(model_checkpoints_path, f"{model_name}_checkpoint_epoch_{epoch}.pth"))
    for old in _epoch_checkpoints(model_checkpoints_path, model_name)[:-3]:
        os.remove(old[1])


def _epoch_checkpoints(path, name):
    pat = re.compile(re.escape(name) + r"_checkpoint_epoch_(\d+)\.pth$")
    found = []
    for f in os.listdir(path) if os.path.isdir(path) else []:
        m = pat.match(f)
        if m:
            found.append((int(m.group(1)), os.path.join(path, f)))
    return sorted(found)


def _dist_rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def load_latest_checkpoint(model, optimizer, scheduler, scaler, model_checkpoints_path, model_name, device,
                           isCheckpoint=True):
    """-> (start_epoch, train_loss | None); newest epoch checkpoint, or the min-loss one (which also
    removes the epoch checkpoints, as upstream does, model_utils.py:113-117).
    Data parallel: every rank loads, but the directory is listed BEFORE anything is removed and only rank 0 removes
    (after a barrier, so no rank is still listing); a second barrier keeps a rank from saving into a directory that is
    being pruned."""
    rank, world = _dist_rank_world()
    epochs = _epoch_checkpoints(model_checkpoints_path, model_name)
    path = None
    if isCheckpoint:
        path = epochs[-1][1] if epochs else None
    else:
        best = os.path.join(model_checkpoints_path, f"{model_name}_min_loss.pth")
        if os.path.exists(best):
            path = best
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if path is not None and rank == 0:
            for _, f in epochs:
                if os.path.exists(f):
                    os.remove(f)
        if world > 1:
            dist.barrier()
    if path is None:
        return 1, None
    ck = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    for obj, key in ((optimizer, "optimizer_state_dict"), (scheduler, "scheduler_state_dict"), (scaler, "scaler_state_dict")):
        if obj is not None and ck.get(key) is not None:
            obj.load_state_dict(ck[key])
    print(f"Resumed from {path} (epoch {ck['epoch']})")
    return ck["epoch"] + 1, ck.get("train_loss")


def custom_collate_fn(batch):
    batch = [b for b in batch if b is not None]
    return torch.utils.data.dataloader.default_collate(batch) if batch else None


# ---------------------------------------------------------------------------------------------- training
_ITERS = {"n": -1}


class DevicePrefetcher:
    """Iterates a loader one batch ahead: batch i+1 travels host -> device on a COPY stream (from pinned memory, so the
    DMA engine moves it) while the compute stream runs step i.  Works for fp32 CHW batches (the reference's transform,
    model_utils.py:539-547) and for uint8 HWC batches (dataset.uint8_hwc: a quarter of the PCIe bytes; ToTensor +
    Normalize then run on the GPU inside frx_input_prep).  Upstream does a pageable synchronous `.to(device)` (:173)."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(self.device)

    def _upload(self, batch):
        if batch is None or batch[0] is None or not self.cuda:
            return batch, None
        with torch.cuda.stream(self.copy_stream):
            moved = []
            for t in batch:
                if isinstance(t, torch.Tensor):
                    if not t.is_cuda and not t.is_pinned():
                        t = t.pin_memory()
                    moved.append(t.to(self.device, non_blocking=True))
                else:
                    moved.append(t)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return tuple(moved), ev

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            if ev is not None:
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ev)
                for t in cur:                      # the caching allocator must not recycle the batch while `main` uses it
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(main)
            yield cur


def _is_plain_ce(criterion):
    return (isinstance(criterion, nn.CrossEntropyLoss) and criterion.weight is None and criterion.reduction == "mean"
            and getattr(criterion, "label_smoothing", 0.0) == 0.0 and criterion.ignore_index == -100)


def train_model(model, train_loader, criterion, optimizer, scaler, device, epoch, epochs, args):
    """One epoch; returns the sample-weighted mean loss (model_utils.py:147-216)."""
    model.train()
    meters = {k: AverageMeter(n, f) for k, n, f in (
        ("bt", "Time", ":6.3f"), ("dt", "Data", ":6.3f"), ("tp", "ThroughPut", ":.2f"), ("loss", "Loss", ":.3f"),
        ("lid", "L_ID", ":.3f"), ("lmag", "L_mag", ":.6f"), ("a1", "Acc@1", ":6.2f"), ("a5", "Acc@5", ":6.2f"))}
    progress = ProgressMeter(len(train_loader) if hasattr(train_loader, "__len__") else 0,
                             [meters["bt"], meters["dt"], meters["tp"], "images/s", meters["loss"],
                              meters["lid"], meters["lmag"], meters["a1"], meters["a5"]],
                             prefix=f"Epoch: [{epoch}/{epochs}]")
    # (loss_g is the integer 0 for every head but MagFace, whose fused backward takes lambda_g itself)
    fused = isinstance(model, NativeFaceNet) and isinstance(optimizer, FusedSGD) and _is_plain_ce(criterion)
    lambda_g = float(getattr(args, "lambda_g", 0.0))
    lead = _dist_rank_world()[0] == 0
    _ITERS["n"] += 1
    pending = []                 # (device loss, device top-k counts, batch size) not yet synced to the host
    end = time.time()

    def flush():
        for loss_t, topk_t, n, lr, lg_t in pending:
            lid = float(loss_t)
            mag = lambda_g * (float(lg_t) if lg_t is not None else 0.0)
            lv = lid + mag
            a1, a5 = (float(v) * 100.0 / n for v in topk_t.tolist())
            meters["loss"].update(lv, n); meters["lid"].update(lid, n); meters["lmag"].update(mag, n)
            meters["a1"].update(a1, n); meters["a5"].update(a5, n)
            if lead:                # (wandb.init ran on rank 0 only: main_pipeline)
                wandb.log({"loss": lv, "loss_id": lid, "loss_mag": mag, "acc1": a1, "acc5": a5, "lr": lr, "epoch": epoch,
                           "step": _ITERS["n"]}, step=_ITERS["n"])
            _ITERS["n"] += 1
        pending.clear()

    for i, batch in enumerate(DevicePrefetcher(train_loader, device)):
        if batch is None or batch[0] is None:
            continue
        images, target = batch
        meters["dt"].update(time.time() - end)
        n = images.size(0)
        lr = optimizer.param_groups[0]["lr"]
        if fused:
            # whole step inside the engine (frx/ddp.py: DataParallelStep): no [N,C] logits, no per-step host sync;
            # from the second step of a batch size on it is a hipGraph replay (plus the gradient all-reduces
            # between the graph segments when the model is a data-parallel replica)
            eng = model._engine_for(n, images.device)
            model._resync_if_touched()
            model._sync_head_flags(eng)
            optimizer._apply_pending(eng)
            if eng.kind == ops.SPHERE:
                eng.sphere_iter = model.head.iter
            eng.set_lambda_g(lambda_g)
            g0 = optimizer.param_groups[0]
            eng.sgd_momentum, eng.sgd_weight_decay = g0["momentum"], g0["weight_decay"]
            out = model._stepper_for(eng, images).step(images, target.contiguous(), lr)
            if eng.kind == ops.SPHERE:
                model.head.iter = eng.sphere_iter
            model._synced_version = model._version_sum()
            pending.append((out["loss"].clone(), out["topk"].clone(), n, lr,
                            out["loss_g"].clone() if eng.kind == ops.MAG else None))
        else:
            output, norm, loss_g, one_hot = model(images, target)
            cosine_s, logits = output
            loss_id = criterion(logits, target)
            loss = loss_id + args.lambda_g * loss_g
            acc1, acc5 = accuracy(cosine_s, target, topk=(1, 5))
            optimizer.zero_grad()
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
            lv, lidv = loss.item(), loss_id.item()
            mag = args.lambda_g * (loss_g.item() if isinstance(loss_g, torch.Tensor) else loss_g)
            meters["loss"].update(lv, n); meters["lid"].update(lidv, n); meters["lmag"].update(mag, n)
            meters["a1"].update(acc1[0].item(), n); meters["a5"].update(acc5[0].item(), n)
            if lead:
                wandb.log({"loss": lv, "loss_id": lidv, "loss_mag": mag, "acc1": acc1[0].item(), "acc5": acc5[0].item(),
                           "lr": lr, "epoch": epoch, "step": _ITERS["n"]}, step=_ITERS["n"])
            _ITERS["n"] += 1
        now = time.time()
        meters["bt"].update(now - end)
        meters["tp"].update(n / max(now - end, 1e-9))
        end = now
        if i % args.print_freq == 0:
            flush()
            progress.display(i)
    flush()
    return meters["loss"].avg


# ---------------------------------------------------------------------------------------------- verification
@torch.no_grad()
def _pair_similarities(model, dataset, batch_size, device):
    """cosine of the two embeddings of every item of a (img1, img2, same) dataset, on the device"""
    model.eval()
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=0)
    sims, labels = [], []
    for img1, img2, same in loader:
        if img1 is None:
            continue
        f1 = model(img1.to(device)).float().contiguous()
        f2 = model(img2.to(device)).float().contiguous()
        sims.append(ops.pair_cosine(f1, f2) if f1.is_cuda else (F.normalize(f1, dim=1) * F.normalize(f2, dim=1)).sum(1))
        labels.append(torch.as_tensor(same).to(device).long())
    if not sims:
        return torch.zeros(0, device=device), torch.zeros(0, dtype=torch.long, device=device)
    return torch.cat(sims), torch.cat(labels)


def _count_correct(cos, same, threshold):
    if cos.is_cuda:
        return int(ops.threshold_count(cos.contiguous(), same.contiguous(), float(threshold)).item())
    return int(((cos > float(np.float32(threshold))).long() == same).sum().item())


def evaluate(model, dataset, batch_size, device, threshold=0.33):
    """accuracy % at a fixed threshold, strict `>` (model_utils.py:354-377)"""
    cos, same = _pair_similarities(model, dataset, batch_size, device)
    return 100.0 * _count_correct(cos, same, threshold) / same.numel() if same.numel() else 0.0


def tune_threshold_roc(model, dataset, batch_size, device):
    """-> (threshold maximising TPR - FPR, accuracy % at it)  (model_utils.py:379-414)"""
    cos, same = _pair_similarities(model, dataset, batch_size, device)
    c, s = cos.cpu().numpy(), same.cpu().numpy()
    thr = V.youden_threshold(s, c)
    return thr, 100.0 * float(((c > thr).astype(int) == s).sum()) / len(s)


def compute_auc(model, dataset, batch_size, device):
    cos, same = _pair_similarities(model, dataset, batch_size, device)
    return V.auc(same.cpu().numpy(), cos.cpu().numpy())


def read_pair_list(pairs_file):
    """whitespace-separated `a b label` integer triples (model_utils.py:423-433)"""
    pairs = []
    with open(pairs_file) as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 3:
                pairs.append((int(parts[0]), int(parts[1]), int(parts[2])))
    return np.asarray(pairs, dtype=np.int64).reshape(-1, 3)


@torch.no_grad()
def embed_ids(model, ids, load_fn, batch_size, device):
    """embeddings [len(ids), 512] of distinct image ids, each image forwarded once"""
    model.eval()
    out = []
    for lo in range(0, len(ids), batch_size):
        batch = torch.stack([load_fn(i) for i in ids[lo:lo + batch_size]]).to(device)
        out.append(model(batch).float())
    return torch.cat(out) if out else torch.zeros(0, 512, device=device)


def kfold_from_similarities(cos, labels, k_fold=10):
    """The shipped protocol on cached similarities (model_utils.py:438-468): StratifiedKFold(k, shuffle, seed 42);
    per fold tune on the held-out part, score accuracy / AUC on the rest; mean and population std."""
    cos_h = cos.detach().cpu().numpy() if isinstance(cos, torch.Tensor) else np.asarray(cos)
    labels = np.asarray(labels)
    folds = V.stratified_folds(labels, k_fold, 42)
    accs, aucs = [], []
    for f in range(k_fold):
        val, trn = folds == f, folds != f
        thr = V.youden_threshold(labels[val], cos_h[val])
        if isinstance(cos, torch.Tensor) and cos.is_cuda:
            idx = torch.from_numpy(np.nonzero(trn)[0]).to(cos.device)
            correct = _count_correct(cos[idx].contiguous(), torch.from_numpy(labels[trn]).to(cos.device), thr)
        else:
            correct = int(((cos_h[trn] > thr).astype(int) == labels[trn]).sum())
        accs.append(100.0 * correct / int(trn.sum()))
        aucs.append(V.auc(labels[trn], cos_h[trn]))
        print(f"=== Fold {f + 1}/{k_fold} ===  threshold {thr:.4f}  accuracy (k-1 folds) {accs[-1]:.3f}%  AUC {aucs[-1]:.4f}")
    return float(np.mean(accs)), float(np.std(accs)), float(np.mean(aucs)), float(np.std(aucs))


def cross_validate_kfold(model, pairs_file, img_dir, transform, device, batch_size=64, k_fold=10):
    """-> (mean_acc, std_acc, mean_auc, std_auc) on an LFW-style pair list (model_utils.py:416-474)"""
    pairs = read_pair_list(pairs_file)
    ds = FlatPairDataset(pairs, img_dir, transform)
    ids = np.unique(pairs[:, :2])
    emb = embed_ids(model, ids.tolist(), ds.load_id, batch_size, device)
    pos = {int(v): i for i, v in enumerate(ids)}
    ia = torch.tensor([pos[int(a)] for a in pairs[:, 0]], device=emb.device)
    ib = torch.tensor([pos[int(b)] for b in pairs[:, 1]], device=emb.device)
    f1, f2 = emb[ia].contiguous(), emb[ib].contiguous()
    cos = ops.pair_cosine(f1, f2) if emb.is_cuda else (F.normalize(f1, dim=1) * F.normalize(f2, dim=1)).sum(1)
    res = kfold_from_similarities(cos, pairs[:, 2], k_fold)
    print(f"\n{k_fold}-fold Results:\nAccuracy: {res[0]:.3f}% ± {res[1]:.3f}%\nAUC:      {res[2]:.4f} ± {res[3]:.4f}")
    return res


# ---------------------------------------------------------------------------------------------- pipeline
def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--batch_size", "-bs", type=int, default=512)
    p.add_argument("--epochs", "-e", type=int, default=30)
    p.add_argument("--learning_rate", "-lr", type=float, default=0.1)
    p.add_argument("--lambda_g", type=float, default=0.0, help="Magnitude loss weight")
    p.add_argument("--print_freq", type=int, default=100)
    p.add_argument("--continue_train", choices=["min_loss", "latest"],
                   help="resume from the best (min_loss) or the newest (latest) checkpoint; default: from scratch")
    p.add_argument("--model-save-path", type=str, default=f"{WORKING_PATH}/models")
    p.add_argument("--wandb-project", type=str, default="face-recognition-training")
    return p.parse_args(argv)


def make_optimizer(model, lr):
    if isinstance(model, NativeFaceNet):
        return FusedSGD(model, lr, momentum=0.9, weight_decay=5e-4)
    return torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4)


def check_label_range(datasets, num_classes):
    """Labels index the head's class axis on the device; a dataset with more identity folders than `num_classes`
    must fail HERE, like upstream's CrossEntropyLoss would on the first such batch (the kernels themselves clamp the
    index and turn the loss into NaN rather than read out of bounds)."""
    for ds in datasets:
        n = getattr(ds, "num_of_identities", None)
        if n is None and hasattr(ds, "samples") and len(ds.samples):
            n = 1 + max(int(lbl) for _, lbl in ds.samples)
        if n is not None and n > num_classes:
            raise ValueError(f"dataset holds {n} identities but the head was built for num_classes={num_classes}: "
                             f"labels would fall outside [0, {num_classes})")


def init_data_parallel():
    """One process per GPU (launched by `python -m torch.distributed.run --nproc-per-node N arcface.py ...`): join the
    RCCL process group.  -> (rank, world, device); (0, 1, default device) when not launched that way.  The reference
    has no multi-GPU path."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or not torch.cuda.is_available():
        return 0, 1, torch.device("cuda" if torch.cuda.is_available() else "cpu")
    import torch.distributed as dist
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return rank, world, device


def main_pipeline(model_class, model_name, project_name, model_final_filename, model_best_filename, num_classes,
                  working_path, dataset_path):
    from torch.utils.data.distributed import DistributedSampler
    from .dataset import uint8_hwc
    t_start = time.time()
    args = parse_args()
    rank, world, device = init_data_parallel()
    lead = rank == 0
    if lead:
        wandb.init(project=project_name, name=model_name, config=vars(args), dir=f"{WORKING_PATH}/wandb")
    ckpt_dir = f"{working_path}/checkpoints/{model_name}"
    if lead and args.continue_train is None and os.path.exists(ckpt_dir):
        shutil.rmtree(ckpt_dir)
        print("Training from scratch, reset all checkpoints...")
    os.makedirs(ckpt_dir, exist_ok=True)
    print(f"Training using {device} - batch size {args.batch_size}{' per GPU x ' + str(world) if world > 1 else ''} - "
          f"epochs {args.epochs} - learning rate {args.learning_rate}")

    model = model_class(num_classes=num_classes, backbone=BACKBONE).to(device)
    native = isinstance(model, NativeFaceNet) and device.type == "cuda"
    root = f"{dataset_path}/CASIA-WebFace"
    # native path: batches stay uint8 HWC until they are on the GPU (a quarter of the PCIe bytes; frx_input_prep applies
    # ToTensor + Normalize there); otherwise the reference's transform (model_utils.py:539-547)
    tf = uint8_hwc if native else None
    parts = [CASIAwebfaceDataset(root, "train", transform=tf), CASIAwebfaceDataset(root, "valid", transform=tf)]
    check_label_range(parts, num_classes)
    train_dataset = ConcatDataset(parts)
    sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None
    train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                              num_workers=8, collate_fn=custom_collate_fn, pin_memory=True, drop_last=world > 1)

    if world > 1 and native:
        model.data_parallel()
    criterion = nn.CrossEntropyLoss().to(device)
    optimizer = make_optimizer(model, args.learning_rate)
    scheduler = get_scheduler(optimizer, "customstep")
    scaler = GradScaler(enabled=False)         # bf16 / fp32 engine: no loss scaling
    start_epoch, best = load_latest_checkpoint(model, optimizer, scheduler, scaler, ckpt_dir, model_name, device,
                                               isCheckpoint=(args.continue_train == "latest"))
    best = float("inf") if best is None else best
    last = args.epochs + start_epoch - 1
    for epoch in range(start_epoch, last + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        loss = train_model(model, train_loader, criterion, optimizer, scaler, device, epoch, last, args)
        if lead and loss < best:
            best = loss
            save_checkpoint(model, optimizer, scheduler, scaler, loss, epoch, ckpt_dir, model_name, isCheckpoint=False)
            print(f"New best model saved: {loss:.6f}")
        if lead:
            save_checkpoint(model, optimizer, scheduler, scaler, loss, epoch, ckpt_dir, model_name, isCheckpoint=True)
        scheduler.step()
    if lead:
        torch.save(model.state_dict(), f"{ckpt_dir}/{model_final_filename}")
        wandb.save(f"{ckpt_dir}/*")
        wandb.finish()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    print(f"Code runs in {time.time() - t_start}s")
