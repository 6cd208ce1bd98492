"""metrics_eval (host side) against a direct sklearn computation on synthetic data."""
import numpy as np
import torch

from aaclip_hip import synth


def test_metrics_eval_matches_sklearn():
    import forward_utils as FU
    from sklearn.metrics import average_precision_score, roc_auc_score
    rng = np.random.default_rng(0)
    masks = synth.synth_masks(6, 64, seed=3).numpy()
    preds = rng.normal(size=(6, 64, 64)) + 2.0 * masks
    labels = np.array([0, 1, 0, 1, 1, 0])
    scores = rng.normal(size=6) + labels
    r = FU.metrics_eval(masks, labels, preds, scores, "x", "Industrial")
    pn = (preds - preds.min()) / (preds.max() - preds.min())
    sn = (scores - scores.min()) / (scores.max() - scores.min())
    assert abs(r["pixel AUC"] - round(roc_auc_score(masks.reshape(-1), pn.reshape(-1)), 4) * 100) < 1e-9
    assert abs(r["pixel AP"] - round(average_precision_score(masks.reshape(-1), pn.reshape(-1)), 4) * 100) < 1e-9
    comb = 0.5 * pn.max(axis=(1, 2)) + 0.5 * sn
    assert abs(r["image AUC"] - round(roc_auc_score(labels, comb), 4) * 100) < 1e-9
    rm = FU.metrics_eval(masks, labels, preds[:, None], scores, "x", "Medical")
    assert abs(rm["image AUC"] - round(roc_auc_score(labels, pn.max(axis=(1, 2))), 4) * 100) < 1e-9
    r0 = FU.metrics_eval(masks, np.zeros(6), preds, scores, "x", "Industrial")
    assert r0["image AUC"] == 0 and r0["image AP"] == 0
