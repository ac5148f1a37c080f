#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the reference never
travels, only these numeric fixtures do.  Absent third-party names the reference
imports at module level (torchvision, wandb, dotenv, alive_progress) are replaced by
empty in-memory stand-ins that are never called on the paths exercised here
(SURVEY.md Appendix C).  Usage:  python tests/golden/make_golden.py
"""
import contextlib
import math
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/main_code"
WANDB_LOG = []


def _mod(name, **a):
    m = types.ModuleType(name)
    m.__dict__.update(a)
    sys.modules[name] = m
    return m


def import_reference():
    names = ["resnet50", "resnet18", "ResNet18_Weights", "ResNet50_Weights", "efficientnet_b0",
             "EfficientNet_B0_Weights", "mobilenet_v2", "MobileNet_V2_Weights"]
    tvm = _mod("torchvision.models", **{n: None for n in names})
    tvt = _mod("torchvision.transforms",
               **{n: (lambda *a, **k: None) for n in ["Compose", "ToTensor", "Normalize", "Resize"]})
    _mod("torchvision", models=tvm, transforms=tvt)
    _mod("wandb", init=lambda **k: None,
         log=lambda d, step=None: WANDB_LOG.append((step, {k: float(v) for k, v in d.items()})),
         save=lambda *a, **k: None, finish=lambda: None)
    _mod("dotenv", load_dotenv=lambda *a, **k: False)
    _mod("alive_progress", alive_bar=contextlib.nullcontext)
    sys.path.insert(0, REF)
    import utils.criterion as RC
    import utils.metrics as RMET
    import utils.model_utils as RM
    import utils.schedulers as RS
    return RC, RM, RMET, RS


def main():
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    RC, RM, RMET, RS = import_reference()
    torch.set_num_threads(4)

    # ------------------------------------------------------------------ heads
    N, D, C = 32, 512, 100          # the D the kernels run at (the K loop of the cosine GEMM is 512 deep)

    def make_head(kind):
        with contextlib.redirect_stdout(None):
            if kind == "arcface":
                return RC.ArcFace(D, C, s=64.0, m=0.5, easy_margin=False), "weight"
            if kind == "arcface_easy":
                return RC.ArcFace(D, C, s=64.0, m=0.5, easy_margin=True), "weight"
            if kind == "sphereface_m4":
                return RC.SphereFace(D, C, m=4), "weight"
            if kind == "cosface":
                return RC.CosFace(D, C, s=64.0, m=0.35), "kernel"
            if kind == "sphereface":
                return RC.SphereFace(D, C, m=2), "weight"
            return RC.CurricularFace(D, C, m=0.5, s=64.0, momentum=0.01), "kernel"

    def make_inputs(head, pname, seed):
        g = torch.Generator().manual_seed(seed)
        w = getattr(head, pname).detach()
        wc = w if pname == "weight" else w.t()          # [C,D]
        y = torch.randint(0, C, (N,), generator=g)
        y[0], y[1] = 0, C - 1                           # label edge values
        x = torch.randn(N, D, generator=g)
        wn = F.normalize(wc, dim=1)
        # rows 2..9: aligned with their class (large target cosine -> margin branch, Curricular mask)
        for i in range(2, 10):
            x[i] = wn[y[i]] * (3.0 + i) + 0.05 * i * torch.randn(D, generator=g)
        x[10] = -4.0 * wn[y[10]]                        # target cosine below th = cos(pi-m)
        x[11] = 2.5 * wn[y[11]]                         # cos == 1 up to rounding (clamp limits)
        x[12] = 5.0 * wn[(y[12] + 1) % C] + 0.3 * wn[y[12]]   # a non-target far above the target
        return x.contiguous(), y

    for kind in ["arcface", "cosface", "sphereface", "curricular", "arcface_easy", "sphereface_m4"]:
        torch.manual_seed({"arcface": 1, "cosface": 2, "sphereface": 3, "curricular": 4, "arcface_easy": 5, "sphereface_m4": 6}[kind])
        head, pname = make_head(kind)
        head.train()
        out = {}
        # state "fresh" = first forward; state "warm" = 4th forward (iter / t have moved)
        for call in range(4):
            x, y = make_inputs(head, pname, 100 + call)
            x.requires_grad_(True)
            p = getattr(head, pname)
            p.grad = None
            pre_t = float(head.t) if hasattr(head, "t") else 0.0
            pre_iter = int(getattr(head, "iter", 0))
            (cos_s, logits), norms, loss_g, one_hot = head(x, y)
            loss = F.cross_entropy(logits, y)
            loss.backward()
            acc1, acc5 = RMET.accuracy(cos_s, y, topk=(1, 5))
            if call in (0, 3):
                tag = "fresh" if call == 0 else "warm"
                out.update({
                    f"{tag}_x": x.detach().numpy(), f"{tag}_y": y.numpy(),
                    f"{tag}_w": p.detach().numpy().copy(),
                    f"{tag}_cos_s": cos_s.detach().numpy(), f"{tag}_logits": logits.detach().numpy(),
                    f"{tag}_norms": norms.detach().numpy(), f"{tag}_loss": np.float64(loss.item()),
                    f"{tag}_dx": x.grad.numpy().copy(), f"{tag}_dw": p.grad.numpy().copy(),
                    f"{tag}_acc1": np.float64(acc1.item()), f"{tag}_acc5": np.float64(acc5.item()),
                    f"{tag}_pre_t": np.float64(pre_t), f"{tag}_pre_iter": np.int64(pre_iter),
                    f"{tag}_post_t": np.float64(float(head.t) if hasattr(head, "t") else 0.0),
                    f"{tag}_post_iter": np.int64(getattr(head, "iter", 0)),
                    f"{tag}_lamb": np.float64(getattr(head, "lamb", 0.0)),
                    f"{tag}_onehot_sum": np.float64(one_hot.sum().item()),
                    f"{tag}_loss_g": np.float64(float(loss_g)),
                })
        np.savez(os.path.join(HERE, f"heads_{kind}.npz"), **out)
        print("heads", kind, "loss fresh/warm", out["fresh_loss"], out["warm_loss"])

    # ------------------------------------------------------------------ CustomStepLR
    lin = nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
    with contextlib.redirect_stdout(None):
        sch = RS.get_scheduler(opt, "customstep")
        lrs = []
        for e in range(70):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
    np.savez(os.path.join(HERE, "customstep_lr.npz"), lrs=np.asarray(lrs, dtype=np.float64))

    # ------------------------------------------------------------------ verify arithmetic
    from sklearn.metrics import roc_auc_score, roc_curve
    from sklearn.model_selection import StratifiedKFold
    from torch.utils.data import TensorDataset

    class Ident(nn.Module):
        def forward(self, x):
            return x

    rng = np.random.RandomState(7)
    ver = {}
    for tag, P, sep in [("a", 600, 0.35), ("b", 257, 0.05), ("c", 64, 1.5)]:
        same = (rng.rand(P) < 0.5).astype(np.int64)
        f1 = rng.randn(P, 32).astype(np.float32)
        f2 = (rng.randn(P, 32) + sep * same[:, None] * f1 * 3).astype(np.float32)
        if tag == "b":                       # exact ties in the similarity values
            f1[10:20] = f1[10]
            f2[10:20] = f2[10]
        ds = TensorDataset(torch.from_numpy(f1), torch.from_numpy(f2), torch.from_numpy(same))
        thr, acc = RM.tune_threshold_roc(Ident(), ds, 64, torch.device("cpu"))
        cos = (F.normalize(torch.from_numpy(f1), dim=1) * F.normalize(torch.from_numpy(f2), dim=1)).sum(1).numpy()
        evals = [float(thr), 0.33, float(cos[3]), float(np.median(cos))]   # cos[3]: tie AT the threshold
        accs = [RM.evaluate(Ident(), ds, 64, torch.device("cpu"), t) for t in evals]
        fpr, tpr, thrs = roc_curve(same, cos)
        ver.update({f"{tag}_f1": f1, f"{tag}_f2": f2, f"{tag}_same": same, f"{tag}_cos": cos,
                    f"{tag}_thr": np.float64(thr), f"{tag}_acc": np.float64(acc),
                    f"{tag}_eval_thr": np.asarray(evals), f"{tag}_eval_acc": np.asarray(accs),
                    f"{tag}_fpr": fpr, f"{tag}_tpr": tpr, f"{tag}_thrs": thrs,
                    f"{tag}_auc": np.float64(roc_auc_score(same, cos))})
    np.savez(os.path.join(HERE, "verify_threshold.npz"), **ver)

    # StratifiedKFold index sets (sklearn itself) for LFW-shaped and ragged label vectors
    kf = {}
    lab_lfw = np.concatenate([np.r_[np.ones(300), np.zeros(300)] for _ in range(10)]).astype(np.int64)
    lab_rag = (rng.rand(1237) < 0.3).astype(np.int64)
    lab_zero_first = np.r_[np.zeros(55), np.ones(45)].astype(np.int64)
    for tag, lab in [("lfw", lab_lfw), ("rag", lab_rag), ("zf", lab_zero_first)]:
        folds = np.empty(len(lab), dtype=np.int32)
        skf = StratifiedKFold(n_splits=10, shuffle=True, random_state=42)
        for f, (_, val) in enumerate(skf.split(np.zeros((len(lab), 3)), lab)):
            folds[val] = f
        kf[f"{tag}_labels"] = lab.astype(np.int8)
        kf[f"{tag}_folds"] = folds.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "verify_kfold_sets.npz"), **kf)

    # cross_validate_kfold end-to-end through the reference.  Two deviations, both recorded in
    # (1) FlatPairDataset reads JPEGs -> replaced by a feature-table dataset with
    # the same (img1, img2, same) contract; (2) roc_auc_score is never imported upstream
    # (NameError) -> injected so the function can complete.
    P = 600
    nid = 400
    table = rng.randn(nid, 48).astype(np.float32)
    a = rng.randint(0, nid, P)
    same = np.r_[np.ones(P // 2), np.zeros(P // 2)].astype(np.int64)
    rng.shuffle(same)
    b = rng.randint(0, nid, P)
    noisy = table + 0.9 * rng.randn(nid, 48).astype(np.float32)

    class TablePairs(torch.utils.data.Dataset):
        def __init__(self, pairs, img_dir, transform):
            self.pairs = pairs

        def __len__(self):
            return len(self.pairs)

        def __getitem__(self, i):
            ai, bi, lab = self.pairs[i]
            f2 = noisy[ai] if lab == 1 else table[bi]
            return torch.from_numpy(table[ai]), torch.from_numpy(f2), int(lab)

    pair_file = os.path.join(HERE, "_pairs_tmp.list")
    with open(pair_file, "w") as f:
        for i in range(P):
            f.write(f"{a[i]} {b[i]} {same[i]}\n")
    RM.FlatPairDataset = TablePairs
    RM.roc_auc_score = roc_auc_score
    with contextlib.redirect_stdout(None):
        res = RM.cross_validate_kfold(Ident(), pair_file, "unused", None, torch.device("cpu"), batch_size=128, k_fold=10)
    os.remove(pair_file)
    f1 = table[a]
    f2 = np.where(same[:, None] == 1, noisy[a], table[b])
    np.savez(os.path.join(HERE, "verify_kfold_e2e.npz"), f1=f1, f2=f2, same=same,
             result=np.asarray(res, dtype=np.float64))
    print("kfold e2e", res)

    # ------------------------------------------------------------------ train_model loop ordering
    class ToyNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = nn.Sequential(nn.Flatten(), nn.Linear(3 * 8 * 8, 64))
            with contextlib.redirect_stdout(None):
                self.arcface = RC.ArcFace(64, 10, s=64.0, m=0.5, easy_margin=False)

        def forward(self, x, labels=None):
            f = self.backbone(x)
            return self.arcface(f, labels) if self.training else f

    torch.manual_seed(11)
    net = ToyNet()
    init = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    batches = [(torch.rand(6, 3, 8, 8, generator=g) * 2 - 1, torch.randint(0, 10, (6,), generator=g)) for _ in range(3)]
    batches.insert(1, (None, None))          # the loop skips empty batches (model_utils.py:169)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
    args = types.SimpleNamespace(lambda_g=0.0, print_freq=1000)
    from torch.amp import GradScaler
    WANDB_LOG.clear()
    with contextlib.redirect_stdout(None):
        avg = RM.train_model(net, batches, nn.CrossEntropyLoss(), opt, GradScaler(), torch.device("cpu"), 1, 1, args)
    tm = {"avg_loss": np.float64(avg)}
    for k, v in init.items():
        tm["init." + k] = v
    for k, v in net.state_dict().items():
        tm["final." + k] = v.detach().numpy()
    real = [b for b in batches if b[0] is not None]
    tm["images"] = np.stack([b[0].numpy() for b in real])
    tm["labels"] = np.stack([b[1].numpy() for b in real])
    for key in ["loss", "loss_id", "loss_mag", "acc1", "acc5", "lr", "epoch", "step"]:
        tm["log_" + key] = np.asarray([d[key] for _, d in WANDB_LOG], dtype=np.float64)
    np.savez(os.path.join(HERE, "train_loop_toy.npz"), **tm)
    print("train loop avg loss", avg, "steps logged", len(WANDB_LOG))


if __name__ == "__main__":
    main()
