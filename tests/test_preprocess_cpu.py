"""CPU tests of the image pre-processing restatement (oracle/preprocess_oracle.py) and of the
library's host-side weight tables.  The transform under test is the reference's
dataset/__init__.py:150-161 (Resize BICUBIC -> ToTensor -> Normalize); its arithmetic is Pillow's,
which is importable, so the oracle is pinned to Pillow itself plus the committed fixtures.
All comparisons are bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def test_oracle_matches_golden(golden):
    for i, (h, w, s) in enumerate(golden["cases"]):
        img = golden[f"in{i}"]
        assert img.shape == (h, w, 3)
        r = P.resize_bicubic_u8(img, int(s))
        assert np.array_equal(r, golden[f"u8_{i}"]), f"case {i}: resize differs"
        assert np.array_equal(P.to_normalised_chw(r), golden[f"f32_{i}"]), f"case {i}: normalise differs"


@pytest.mark.parametrize("h,w,s", [(1024, 1024, 518), (700, 700, 518), (900, 840, 518), (256, 300, 518),
                                   (518, 518, 518), (1000, 518, 518), (37, 91, 70), (5, 3, 70), (1, 1, 4)])
def test_oracle_matches_pillow(h, w, s):
    from PIL import Image
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((s, s), Image.BICUBIC))
    assert np.array_equal(P.resize_bicubic_u8(img, s), ref)


def test_extreme_values_clip():
    # saturated checkerboards drive the negative lobes past [0,255]: clipping must match Pillow
    from PIL import Image
    img = np.zeros((64, 64, 3), np.uint8)
    img[::2, ::2] = 255
    img[1::2, 1::2] = 255
    for s in (40, 70, 200):
        ref = np.asarray(Image.fromarray(img).resize((s, s), Image.BICUBIC))
        assert np.array_equal(P.resize_bicubic_u8(img, s), ref)


def test_normalise_is_torch_ops():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    t = torch.from_numpy(img).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    t.sub_(torch.tensor(P.CLIP_MEAN).view(-1, 1, 1)).div_(torch.tensor(P.CLIP_STD).view(-1, 1, 1))
    assert np.array_equal(P.to_normalised_chw(img), t.numpy())


@pytest.mark.parametrize("a,b", [(1024, 518), (700, 518), (256, 518), (37, 70), (2000, 518), (900, 70), (3, 70)])
def test_library_tables_match_oracle(a, b):
    from aaclip_hip import engine
    bounds, coefs = engine.resample_table(a, b)
    k, ob, oc = P.resample_table(a, b)
    assert coefs.shape == (b, k)
    assert np.array_equal(bounds.numpy(), ob)
    assert np.array_equal(coefs.numpy(), oc)


def test_library_identity_table():
    from aaclip_hip import engine
    bounds, coefs = engine.resample_table(70, 70)
    assert coefs.shape == (70, 1) and int(coefs.min()) == 1 << 22 == int(coefs.max())
    assert np.array_equal(bounds[:, 0].numpy(), np.arange(70)) and int(bounds[:, 1].max()) == 1


def test_normalise_lut_matches_oracle():
    from aaclip_hip import engine
    lut = engine._normalise_lut(engine.CLIP_MEAN, engine.CLIP_STD).numpy()
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)   # [1,256,3]
    assert np.array_equal(lut, P.to_normalised_chw(ramp)[:, 0, :])
