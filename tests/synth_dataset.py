"""Writes a small MVTec-AD style tree (PNG images + defect masks) for the harness tests."""
import os

import numpy as np
from PIL import Image


def write_tree(root, classes=("bottle", "grid"), n_good=3, n_bad=3, size=(96, 96), seed=111):
    rng = np.random.default_rng(seed)
    h, w = size
    for c in classes:
        for kind, n in (("good", n_good), ("broken", n_bad)):
            os.makedirs(os.path.join(root, c, "test", kind), exist_ok=True)
            if kind != "good":
                os.makedirs(os.path.join(root, c, "ground_truth", kind), exist_ok=True)
            for i in range(n):
                yy, xx = np.mgrid[0:h, 0:w]
                base = 128 + 60 * np.sin(xx / (5.0 + i) + rng.uniform(0, 3)) * np.cos(yy / (7.0 + i))
                img = np.clip(base[..., None] + rng.integers(-40, 40, (h, w, 3)), 0, 255).astype(np.uint8)
                if kind != "good":
                    y0, x0 = rng.integers(5, h // 2), rng.integers(5, w // 2)
                    hh, ww = rng.integers(8, h // 3), rng.integers(8, w // 3)
                    img[y0:y0 + hh, x0:x0 + ww] = rng.integers(0, 256, (hh, ww, 3))
                    mask = np.zeros((h, w), np.uint8)
                    mask[y0:y0 + hh, x0:x0 + ww] = 255
                    Image.fromarray(mask).save(os.path.join(root, c, "ground_truth", kind, f"{i:03d}_mask.png"))
                Image.fromarray(img).save(os.path.join(root, c, "test", kind, f"{i:03d}.png"))
    return root
