"""Test-time datasets with the reference's names and item layout (reference
dataset/__init__.py:123-272: BaseSingleClassDataset, get_dataset stage "test"/"visualize").

Only the inference side is provided: the training datasets (random rotations / colour jitter,
reference dataset/__init__.py:13-121) belong to the training loop, which is outside the path this
package replaces.  torchvision is not needed: the transform is spelled out (Pillow resize, then the
three tensor operations of ToTensor / Normalize).

`device_preprocess=True` returns the decoded image as uint8 [H,W,3] instead, so that the caller
runs `aaclip_hip.engine.preprocess` on the GPU (bit-identical result, see tests); images of one
class must then share a size for the default DataLoader collate (true for MVTec-AD).
The reference also attaches a random normal "prompt_image" to anomalous samples
(dataset/__init__.py:196-203); only its IQM branch reads it, so it is not produced here.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from .constants import CLASS_NAMES, DATA_PATH, DOMAINS  # noqa: F401

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
METADATA_ROOT = os.environ.get("AACLIP_METADATA_ROOT", "./dataset/metadata")


def transform_image(img: Image.Image, img_size: int) -> torch.Tensor:
    """Resize((S,S), BICUBIC) -> ToTensor -> Normalize (reference dataset/__init__.py:150-161)."""
    arr = np.asarray(img.resize((img_size, img_size), Image.BICUBIC))
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return t.sub_(torch.tensor(CLIP_MEAN).view(-1, 1, 1)).div_(torch.tensor(CLIP_STD).view(-1, 1, 1))


def transform_mask(mask: Image.Image, img_size: int) -> torch.Tensor:
    """Resize NEAREST -> ToTensor -> (!= 0) (reference dataset/__init__.py:162-167,186-189) -> [1,S,S]."""
    arr = np.asarray(mask.resize((img_size, img_size), Image.NEAREST))
    return (torch.from_numpy(arr.copy()).to(torch.float32).div(255).unsqueeze(0) != 0).float()


class BaseSingleClassDataset(Dataset):
    def __init__(self, data_path: str, meta_path: str, img_size: int, class_name: str, logger=None, shot: int = -1,
                 device_preprocess: bool = False):
        assert class_name is not None, "class_name should be provided"
        self.data_path = data_path
        self.img_size = img_size
        self.shot = shot
        self.device_preprocess = device_preprocess
        self.full_shot = "full-shot" in meta_path
        self.meta, self.normal_meta = [], []
        with open(meta_path, "r") as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                m = json.loads(line)
                if m["class_name"] == class_name:
                    self.meta.append(m)
                    if m["label"] == 0:
                        self.normal_meta.append(m)
        if logger:
            logger.info(f"Class name: {class_name}")
            logger.info(f"Sample number: {len(self.meta)}")
            logger.info("=====================================")

    def __len__(self):
        return len(self.meta)

    def __getitem__(self, idx):
        meta = self.meta[idx]
        img = Image.open(os.path.join(self.data_path, meta["image_path"])).convert("RGB")
        if self.device_preprocess:
            image = torch.from_numpy(np.asarray(img).copy())
        else:
            image = transform_image(img, self.img_size)
        if meta["label"]:
            mask = transform_mask(Image.open(os.path.join(self.data_path, meta["mask_path"])).convert("L"),
                                  self.img_size)
        else:
            mask = torch.zeros([1, self.img_size, self.img_size])
        return {"image": image, "mask": mask, "label": meta["label"], "file_name": meta["image_path"],
                "class_name": meta["class_name"]}


def get_dataset(dataset_name: str, img_size: int, training_mode: Optional[str], shot: int = -1, stage: str = "train",
                logger=None, device_preprocess: bool = False) -> Dict[str, BaseSingleClassDataset]:
    """reference dataset/__init__.py:208-272; stages "test" and "visualize" (one dataset per class)."""
    if "Med" not in dataset_name:
        assert dataset_name in DATA_PATH, (
            f"Dataset {dataset_name} not found; available datasets: {list(DATA_PATH.keys())}")
    if stage not in ("test", "visualize"):
        raise NotImplementedError(
            "only the inference datasets are part of the MI355X path; training datasets (stage='train') are not")
    meta_path = os.path.join(METADATA_ROOT, dataset_name, "full-shot.jsonl")
    return {c: BaseSingleClassDataset(DATA_PATH[dataset_name], meta_path, img_size, c, logger=logger, shot=shot,
                                      device_preprocess=device_preprocess)
            for c in CLASS_NAMES[dataset_name]}


def build_metadata(data_path: str, out_path: str, class_names=None) -> int:
    """Write a full-shot.jsonl for an MVTec-AD style tree (<class>/test/<defect>/*.png with masks under
    <class>/ground_truth/<defect>/<stem>_mask.png; 'good' = normal).  Returns the number of rows.
    (The reference ships pre-made metadata files; this regenerates the same row format.)"""
    rows = []
    for c in sorted(class_names or os.listdir(data_path)):
        tdir = os.path.join(data_path, c, "test")
        if not os.path.isdir(tdir):
            continue
        for defect in sorted(os.listdir(tdir)):
            for fn in sorted(os.listdir(os.path.join(tdir, defect))):
                stem, ext = os.path.splitext(fn)
                if ext.lower() not in (".png", ".jpg", ".jpeg", ".bmp"):
                    continue
                good = defect == "good"
                rows.append({"image_path": f"{c}/test/{defect}/{fn}", "label": 0 if good else 1,
                             "mask_path": "" if good else f"{c}/ground_truth/{defect}/{stem}_mask.png",
                             "class_name": c})
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    return len(rows)
