"""CPU restatement of the image pre-processing that feeds the path (TEST INFRASTRUCTURE ONLY).

Reference call site: dataset/__init__.py:150-161 (BaseSingleClassDataset.transform_x) and :62-71
(BaseDataset): `transforms.Resize((S, S), Image.BICUBIC)` on a PIL RGB image, `ToTensor()`,
`Normalize(mean, std)`.  torchvision's Resize on a PIL image is `PIL.Image.resize(size, BICUBIC)`,
so the arithmetic lives in Pillow (third-party, not under /root/reference; requirements do not
pin it, the container has Pillow 12.2.0).  Pillow's published 8-bit resampler, restated here:

  * two passes, horizontal then vertical, each skipped when the size does not change;
    the intermediate image is rounded and clipped to uint8 between the passes;
  * per output index: centre = (i + 0.5) * scale, support = 2 * max(scale, 1) (bicubic, a = -0.5),
    window [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image,
    weights w((x - centre + 0.5) / max(scale, 1)) normalised by their sum in double precision;
  * weights rounded to 22-bit fixed point (round half away from zero), accumulated in int32 on top
    of 1 << 21, shifted right by 22 and clipped to [0, 255].

Pinned by tests/test_preprocess_cpu.py against PIL.Image.resize itself (bit-exact) and the
committed fixtures in tests/golden/preprocess.npz.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_weight(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_table(in_size: int, out_size: int):
    """-> (ksize, bounds int32 [out,2] (first, count), coefs int32 [out, ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coefs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for i in range(out_size):
        center = (i + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = [bicubic_weight((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            coefs[i, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[i] = (xmin, n)
    return ksize, bounds, coefs


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    ksize, bounds, coefs = resample_table(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for i in range(out_size):
        first, n = bounds[i]
        acc = np.tensordot(coefs[i, :n].astype(np.int64), src[first:first + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, size: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [size, size, C], Pillow's BICUBIC resize."""
    assert img.dtype == np.uint8 and img.ndim == 3
    return _pass(_pass(img, size, 1), size, 0)


def to_normalised_chw(img_u8: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """ToTensor + Normalize in fp32: (v / 255 - mean) / std, uint8 [H,W,3] -> float32 [3,H,W]."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)


def preprocess(img_u8: np.ndarray, size: int) -> np.ndarray:
    return to_normalised_chw(resize_bicubic_u8(img_u8, size))
