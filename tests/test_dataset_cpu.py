"""CPU tests of the test-time dataset mirror (reference dataset/__init__.py:123-272)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

import dataset as D
from oracle import preprocess_oracle as P
from synth_dataset import write_tree


@pytest.fixture()
def tree(tmp_path):
    root = write_tree(str(tmp_path / "MVTec"))
    meta = str(tmp_path / "meta" / "MVTec" / "full-shot.jsonl")
    n = D.build_metadata(root, meta)
    assert n == 12
    return root, meta


def test_metadata_rows_have_the_reference_format(tree):
    root, meta = tree
    rows = [json.loads(l) for l in open(meta)]
    assert set(rows[0]) == {"image_path", "label", "mask_path", "class_name"}
    bad = [r for r in rows if r["label"] == 1]
    assert len(bad) == 6 and all(os.path.exists(os.path.join(root, r["mask_path"])) for r in bad)
    assert all(r["mask_path"] == "" for r in rows if r["label"] == 0)


def test_items_match_the_reference_transform(tree):
    root, meta = tree
    ds = D.BaseSingleClassDataset(root, meta, 70, "bottle")
    assert len(ds) == 6 and len(ds.normal_meta) == 3
    for i in range(len(ds)):
        it = ds[i]
        raw = np.asarray(Image.open(os.path.join(root, it["file_name"])).convert("RGB"))
        assert it["image"].dtype == torch.float32 and tuple(it["image"].shape) == (3, 70, 70)
        assert np.array_equal(it["image"].numpy(), P.preprocess(raw, 70))          # bit-exact
        assert tuple(it["mask"].shape) == (1, 70, 70) and set(np.unique(it["mask"].numpy())) <= {0.0, 1.0}
        assert (it["mask"].sum() > 0) == bool(it["label"])
        assert it["class_name"] == "bottle"


def test_mask_is_pillow_nearest(tree):
    root, meta = tree
    ds = D.BaseSingleClassDataset(root, meta, 70, "grid")
    it = next(ds[i] for i in range(len(ds)) if ds[i]["label"])
    m = Image.open(os.path.join(root, it["file_name"].replace("test", "ground_truth").replace(".png", "_mask.png")))
    want = (np.asarray(m.convert("L").resize((70, 70), Image.NEAREST)) != 0).astype(np.float32)
    assert np.array_equal(it["mask"][0].numpy(), want)


def test_device_preprocess_returns_raw_frames(tree):
    root, meta = tree
    ds = D.BaseSingleClassDataset(root, meta, 70, "bottle", device_preprocess=True)
    it = ds[0]
    assert it["image"].dtype == torch.uint8 and tuple(it["image"].shape) == (96, 96, 3)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4)))
    assert tuple(batch["image"].shape) == (4, 96, 96, 3) and tuple(batch["mask"].shape) == (4, 1, 70, 70)


def test_get_dataset_contract(tree, monkeypatch):
    root, meta = tree
    monkeypatch.setattr(D, "METADATA_ROOT", os.path.dirname(os.path.dirname(meta)))
    monkeypatch.setitem(D.DATA_PATH, "MVTec", root)
    sets = D.get_dataset("MVTec", 70, None, -1, "test")
    assert list(sets) == D.CLASS_NAMES["MVTec"] and len(sets["bottle"]) == 6 and len(sets["zipper"]) == 0
    with pytest.raises(AssertionError):
        D.get_dataset("NoSuchSet", 70, None, -1, "test")
    with pytest.raises(NotImplementedError):
        D.get_dataset("MVTec", 70, "full_shot", -1, "train")
