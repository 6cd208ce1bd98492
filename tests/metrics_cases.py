"""Inputs of the metrics_eval fixtures (tests/golden/metrics.npz): the variants derived from one stored base set.
Shared by tests/golden/make_golden_metrics.py (which feeds them to the reference's metrics_eval) and
tests/test_metrics_cpu.py (which feeds them to the build's)."""
import numpy as np

KEYS = ("pixel AUC", "pixel AP", "image AUC", "image AP")


def make_base(seed=20240111, n=12, s=48):
    rng = np.random.default_rng(seed)
    masks = np.zeros((n, s, s), dtype=np.float32)
    labels = np.zeros(n, dtype=np.int64)
    for i in range(n):
        if i % 3 != 0:
            y, x = rng.integers(0, s - 12, size=2)
            h, w = rng.integers(4, 12, size=2)
            masks[i, y:y + h, x:x + w] = 1
            labels[i] = 1
    preds = (rng.normal(size=(n, s, s)) + 1.5 * masks).astype(np.float32)
    scores = (rng.normal(size=n) + 1.2 * labels).astype(np.float32)
    return masks, labels, preds, scores


def derive_cases(masks, labels, preds, scores):
    """name -> (pixel_label, image_label, pixel_preds, image_preds, domain); float32 like test_last.py's arrays."""
    n = preds.shape[0]
    two = np.stack([scores, -scores], axis=1)
    coarse = (np.round(preds * 2) / 2).astype(np.float32)            # many ties: rank handling must agree
    return {
        "industrial": (masks, labels, preds, scores, "Industrial"),
        "medical": (masks, labels, preds, scores, "Medical"),
        "preds_4d": (masks, labels, preds[:, None], scores, "Industrial"),
        "preds_flat": (masks, labels, preds.reshape(n, -1), scores, "Industrial"),
        "image_preds_n2": (masks, labels, preds, two, "Industrial"),
        "ties": (masks, labels, coarse, np.round(scores), "Industrial"),
        "one_image_class": (masks, np.ones(n, dtype=np.int64), preds, scores, "Industrial"),
        "already_unit_max": (masks, labels, (preds - preds.min()) / (preds.max() - preds.min()),
                             (scores - scores.min()) / (scores.max() - scores.min()), "Medical"),
    }
