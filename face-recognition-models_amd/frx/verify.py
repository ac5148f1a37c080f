"""Host-side arithmetic of LFW-style verification: ROC / Youden threshold, stratified folds, AUC.
Device work (embeddings, pair cosine, threshold counts) is in libfrx; what remains here is O(P log P)
bookkeeping on <= a few thousand similarities, restated from scikit-learn's published algorithms so
the product needs no sklearn (reference: main_code/utils/model_utils.py:320-474)."""
from __future__ import annotations

import numpy as np


def roc_points(labels, scores):
    """fpr, tpr, thresholds as sklearn.metrics.roc_curve(labels, scores) returns them (positive label 1,
    collinear interior points dropped, leading (0, 0, inf) point)."""
    labels = np.asarray(labels)
    scores = np.asarray(scores)
    order = np.argsort(-scores, kind="mergesort")          # descending, stable
    sc = scores[order]
    hit = (labels[order] == 1).astype(np.float64)
    last_of_run = np.r_[np.nonzero(np.diff(sc))[0], sc.size - 1]
    tp = np.cumsum(hit)[last_of_run]
    fp = 1.0 + last_of_run - tp
    thr = sc[last_of_run]
    if tp.size > 2:
        corner = np.r_[True, (np.diff(fp, 2) != 0) | (np.diff(tp, 2) != 0), True]
        tp, fp, thr = tp[corner], fp[corner], thr[corner]
    tp, fp, thr = np.r_[0.0, tp], np.r_[0.0, fp], np.r_[np.inf, thr]
    with np.errstate(invalid="ignore", divide="ignore"):
        return fp / fp[-1], tp / tp[-1], thr


def youden_threshold(labels, scores):
    fpr, tpr, thr = roc_points(labels, scores)
    return thr[int(np.argmax(tpr - fpr))]


def auc(labels, scores):
    if np.unique(np.asarray(labels)).size < 2:
        return 0.0
    fpr, tpr, _ = roc_points(labels, scores)
    return float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) * 0.5))


def stratified_folds(labels, n_splits=10, seed=42):
    """fold id per sample, identical to StratifiedKFold(n_splits, shuffle=True, random_state=seed)."""
    y = np.asarray(labels)
    _, first, inv = np.unique(y, return_index=True, return_inverse=True)
    rank_of_class = np.argsort(np.argsort(first))          # classes numbered by first appearance
    enc = rank_of_class[inv]
    n_cls = first.size
    srt = np.sort(enc)
    per_fold = np.stack([np.bincount(srt[i::n_splits], minlength=n_cls) for i in range(n_splits)])
    rng = np.random.RandomState(seed)
    out = np.empty(y.size, dtype=np.int32)
    for k in range(n_cls):
        ids = np.repeat(np.arange(n_splits), per_fold[:, k])
        rng.shuffle(ids)
        out[enc == k] = ids
    return out
