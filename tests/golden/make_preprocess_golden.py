"""Generates tests/golden/preprocess.npz: inputs and outputs of the reference's image transform.

The reference transform (reference dataset/__init__.py:150-161) is torchvision's
Resize((S,S), BICUBIC) -> ToTensor -> Normalize on a PIL image; torchvision is not installed in
this image, but its Resize on a PIL image is `PIL.Image.resize`, and ToTensor/Normalize are the
three torch ops spelled out below, so the vectors come from Pillow + torch directly.
Run from the repo root:  python tests/golden/make_preprocess_golden.py
"""
import os

import numpy as np
import torch
from PIL import Image

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)
CASES = [(96, 96, 70), (150, 130, 70), (40, 56, 70), (70, 70, 70), (70, 100, 70), (301, 70, 70)]


def transform(img_u8, size):
    pil = Image.fromarray(img_u8).resize((size, size), Image.BICUBIC)
    resized = np.asarray(pil).copy()
    t = torch.from_numpy(resized).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    t.sub_(torch.tensor(MEAN).view(-1, 1, 1)).div_(torch.tensor(STD).view(-1, 1, 1))
    return resized, t.numpy()


def main():
    rng = np.random.default_rng(111)
    out = {}
    for i, (h, w, s) in enumerate(CASES):
        # smooth structure + noise so both the negative lobes and the clipping are exercised
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 120 * np.sin(xx / 7.0 + i)[..., None] * np.cos(yy / 5.0)[..., None]
        img = np.clip(base + rng.integers(-90, 90, (h, w, 3)), 0, 255).astype(np.uint8)
        img[: h // 4, : w // 4] = 255
        img[-h // 4:, -w // 4:] = 0
        resized, norm = transform(img, s)
        out[f"in{i}"] = img
        out[f"u8_{i}"] = resized
        out[f"f32_{i}"] = norm
    out["cases"] = np.asarray(CASES, np.int32)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "preprocess.npz"), **out)
    print("pillow", Image.__version__ if hasattr(Image, "__version__") else "", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
