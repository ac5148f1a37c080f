"""Oracle: LFW-style pair verification arithmetic (threshold, accuracy, 10-fold).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/main_code/utils/model_utils.py:
  evaluate              :354-377   pred = cos > thr (strict); acc %
  tune_threshold_roc    :379-414   sklearn roc_curve -> argmax(tpr - fpr)
  compute_auc           :320-352   roc_auc_score (never imported upstream -> NameError)
  cross_validate_kfold  :416-474   StratifiedKFold(10, shuffle=True, random_state=42);
                                   threshold tuned on the held-out fold, accuracy on the
                                   other nine; np.mean / np.std (population)
The sklearn pieces (third-party: scikit-learn, unpinned upstream; 1.7.2 in this image)
are restated from their published algorithms and pinned by tests/golden/verify_*.npz,
which were produced by sklearn itself.
"""
from __future__ import annotations

import numpy as np


def pair_cosine(f1, f2, dtype=np.float32):
    """F.normalize(f1,dim=1) . F.normalize(f2,dim=1) row-wise (model_utils.py:370-372)."""
    f1 = np.asarray(f1, dtype=dtype)
    f2 = np.asarray(f2, dtype=dtype)
    n1 = np.maximum(np.sqrt((f1 * f1).sum(1, keepdims=True)), dtype(1e-12))
    n2 = np.maximum(np.sqrt((f2 * f2).sum(1, keepdims=True)), dtype(1e-12))
    return ((f1 / n1) * (f2 / n2)).sum(1)


def roc_curve(y_true, y_score):
    """sklearn.metrics.roc_curve(y_true, y_score) with defaults (drop_intermediate=True)."""
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score)
    pos = (y_true == 1)
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score = y_score[order]
    pos = pos[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, pos.size - 1]
    tps = np.cumsum(pos.astype(np.float64))[idx]
    fps = 1 + idx - tps
    thr = y_score[idx]
    if len(fps) > 2:  # drop collinear interior points
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    thr = np.r_[np.inf, thr]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.full(fps.shape, np.nan)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.full(tps.shape, np.nan)
    return fpr, tpr, thr


def roc_auc(y_true, y_score):
    """sklearn.metrics.roc_auc_score for binary labels (trapezoid over the full ROC)."""
    fpr, tpr, _ = roc_curve(y_true, y_score)
    return float(np.trapezoid(tpr, fpr))


def tune_threshold_roc(cos, same):
    """model_utils.py:406-412 on precomputed similarities."""
    cos = np.asarray(cos)
    same = np.asarray(same)
    fpr, tpr, thr = roc_curve(same, cos)
    best = thr[int(np.argmax(tpr - fpr))]
    pred = (cos > best).astype(int)
    acc = 100.0 * (pred == same).sum() / len(same)
    return best, acc


def evaluate(cos, same, threshold=0.33):
    """model_utils.py:373-377 on precomputed similarities."""
    cos = np.asarray(cos)
    same = np.asarray(same)
    if len(same) == 0:
        return 0.0
    return 100.0 * int(((cos > threshold).astype(np.int64) == same).sum()) / len(same)


def stratified_kfold_test_folds(labels, n_splits=10, seed=42):
    """sklearn StratifiedKFold(n_splits, shuffle=True, random_state=seed)._make_test_folds.
    Returns test_folds[i] = fold in which sample i is held out."""
    y = np.asarray(labels)
    _, y_idx, y_inv = np.unique(y, return_index=True, return_inverse=True)
    _, class_perm = np.unique(y_idx, return_inverse=True)
    y_enc = class_perm[y_inv]
    n_classes = len(y_idx)
    y_order = np.sort(y_enc)
    allocation = np.asarray([np.bincount(y_order[i::n_splits], minlength=n_classes)
                             for i in range(n_splits)])
    rng = np.random.RandomState(seed)
    test_folds = np.empty(len(y), dtype="i")
    for k in range(n_classes):
        folds_for_class = np.arange(n_splits).repeat(allocation[:, k])
        rng.shuffle(folds_for_class)
        test_folds[y_enc == k] = folds_for_class
    return test_folds


def cross_validate_kfold(cos, same, k_fold=10, with_auc=True):
    """model_utils.py:438-468 on precomputed similarities (one per pair-list line)."""
    cos = np.asarray(cos)
    same = np.asarray(same)
    folds = stratified_kfold_test_folds(same, k_fold, 42)
    accs, aucs, thrs = [], [], []
    for f in range(k_fold):
        val = np.where(folds == f)[0]
        trn = np.where(folds != f)[0]
        thr, _ = tune_threshold_roc(cos[val], same[val])
        accs.append(evaluate(cos[trn], same[trn], thr))
        thrs.append(thr)
        if with_auc:
            aucs.append(roc_auc(same[trn], cos[trn]) if len(np.unique(same[trn])) > 1 else 0.0)
    res = (float(np.mean(accs)), float(np.std(accs)),
           float(np.mean(aucs)) if with_auc else 0.0, float(np.std(aucs)) if with_auc else 0.0)
    return res, np.asarray(accs), np.asarray(thrs)
