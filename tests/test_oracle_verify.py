"""Pin oracle/verify.py to vectors produced by the reference's tune_threshold_roc /
evaluate / cross_validate_kfold and by sklearn (roc_curve, StratifiedKFold).  CPU only."""
import os

import numpy as np
import pytest

from oracle import verify as V


@pytest.fixture(scope="module")
def thr(golden_dir):
    return np.load(os.path.join(golden_dir, "verify_threshold.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_threshold_and_accuracy(thr, tag):
    cos = V.pair_cosine(thr[f"{tag}_f1"], thr[f"{tag}_f2"])
    np.testing.assert_allclose(cos, thr[f"{tag}_cos"], atol=2e-6)
    # feed the reference's own similarities so ties/threshold picks are bit-identical
    best, acc = V.tune_threshold_roc(thr[f"{tag}_cos"], thr[f"{tag}_same"])
    assert best == pytest.approx(float(thr[f"{tag}_thr"]), abs=0)
    assert acc == pytest.approx(float(thr[f"{tag}_acc"]), abs=1e-9)
    for t, a in zip(thr[f"{tag}_eval_thr"], thr[f"{tag}_eval_acc"]):
        assert V.evaluate(thr[f"{tag}_cos"], thr[f"{tag}_same"], t) == pytest.approx(a, abs=1e-9)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_roc_curve_matches_sklearn(thr, tag):
    fpr, tpr, ths = V.roc_curve(thr[f"{tag}_same"], thr[f"{tag}_cos"])
    np.testing.assert_array_equal(fpr, thr[f"{tag}_fpr"])
    np.testing.assert_array_equal(tpr, thr[f"{tag}_tpr"])
    np.testing.assert_array_equal(ths, thr[f"{tag}_thrs"])
    assert V.roc_auc(thr[f"{tag}_same"], thr[f"{tag}_cos"]) == pytest.approx(float(thr[f"{tag}_auc"]), abs=1e-12)


@pytest.mark.parametrize("tag", ["lfw", "rag", "zf"])
def test_stratified_kfold_sets(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "verify_kfold_sets.npz"))
    folds = V.stratified_kfold_test_folds(g[f"{tag}_labels"].astype(np.int64), 10, 42)
    np.testing.assert_array_equal(folds, g[f"{tag}_folds"].astype(np.int32))


def test_cross_validate_kfold_end_to_end(golden_dir):
    g = np.load(os.path.join(golden_dir, "verify_kfold_e2e.npz"))
    cos = V.pair_cosine(g["f1"], g["f2"])
    (mean_acc, std_acc, mean_auc, std_auc), accs, _ = V.cross_validate_kfold(cos, g["same"], 10)
    ref = g["result"]
    # +-0.2 % is the north-star bar; the arithmetic itself should agree to a pair
    assert mean_acc == pytest.approx(ref[0], abs=100.0 / 540 / 10 + 1e-9)
    assert std_acc == pytest.approx(ref[1], abs=0.05)
    assert mean_auc == pytest.approx(ref[2], abs=1e-6)
    assert std_auc == pytest.approx(ref[3], abs=1e-6)
    assert len(accs) == 10


def test_evaluate_edge_cases():
    assert V.evaluate(np.array([]), np.array([]), 0.3) == 0.0
    # strict '>' : a similarity equal to the threshold predicts "different"
    assert V.evaluate(np.array([0.5]), np.array([1]), 0.5) == 0.0
    assert V.evaluate(np.array([0.5]), np.array([0]), 0.5) == 100.0
