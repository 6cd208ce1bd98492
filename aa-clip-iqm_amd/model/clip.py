"""Model factory with the reference's signature (reference model/clip.py:84-202).

Weights come from a local OpenAI state-dict file (`model/ViT-L-14-336px.pt`,
reference model/clip.py:16) when `pretrained` is set; there is no hub access.
`precision` selects the arithmetic type of the matrix products on the MI355X:
'fp32' = exact fp32 MFMA, 'fp16' / 'bf16' = 16-bit MFMA operands with fp32
accumulation, fp32 LayerNorm/softmax statistics and an fp32 residual stream.
Parameters are always kept in fp32.
"""
from __future__ import annotations

import json
import logging
import os
from copy import deepcopy
from pathlib import Path
from typing import Optional, Tuple, Union

import torch

from .model import CLIP, get_cast_dtype, resize_pos_embed
from .transformer import set_precision

_MODEL_CONFIG_PATHS = [Path(__file__).parent / "model_configs/"]
_MODEL_CONFIGS = {}
_MODEL_CKPT_PATHS = {"ViT-L-14-336": Path(__file__).parent / "ViT-L-14-336px.pt"}


def _rescan_model_configs():
    global _MODEL_CONFIGS
    for d in _MODEL_CONFIG_PATHS:
        for cf in sorted(Path(d).glob("*.json")):
            with open(cf) as f:
                cfg = json.load(f)
            if all(k in cfg for k in ("embed_dim", "vision_cfg", "text_cfg")):
                _MODEL_CONFIGS[cf.stem] = cfg


_rescan_model_configs()


def list_models():
    return list(_MODEL_CONFIGS.keys())


def get_model_config(model_name):
    return deepcopy(_MODEL_CONFIGS[model_name]) if model_name in _MODEL_CONFIGS else None


def load_state_dict(checkpoint_path: str, map_location="cpu"):
    """reference model/clip.py:62-70; only weights are unpickled."""
    try:
        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=True)
    except RuntimeError as e:
        if "TorchScript" not in str(e):
            raise
        # OpenAI's released .pt files are TorchScript archives (reference model/openai.py:56-60 tries
        # torch.jit.load first); only the parameters are taken from it.
        ckpt = torch.jit.load(checkpoint_path, map_location="cpu").eval().state_dict()
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    if next(iter(sd.items()))[0].startswith("module"):
        sd = {k[7:]: v for k, v in sd.items()}
    return sd


def _clean_openai_state_dict(sd: dict) -> dict:
    sd = {k: (v.float() if torch.is_floating_point(v) else v) for k, v in sd.items()}
    for k in ("input_resolution", "context_length", "vocab_size"):
        sd.pop(k, None)
    return sd


def load_checkpoint(model, checkpoint_path, strict=True):
    sd = _clean_openai_state_dict(load_state_dict(checkpoint_path))
    resize_pos_embed(sd, model)
    return model.load_state_dict(sd, strict=strict)


def create_model(model_name: str, img_size: int, pretrained: Optional[str] = None, precision: str = "fp32",
                 device: Union[str, torch.device] = "cpu", jit: bool = False, force_quick_gelu: bool = False,
                 force_custom_text: bool = False, force_patch_dropout: Optional[float] = None,
                 force_image_size: Optional[Union[int, Tuple[int, int]]] = None, output_dict: Optional[bool] = None,
                 require_pretrained: bool = False, adapter=False):
    model_name = model_name.replace("/", "-")
    if isinstance(device, str):
        device = torch.device(device)
    if jit:
        raise NotImplementedError("TorchScript is not part of the HIP path")
    if force_custom_text:
        raise NotImplementedError("CustomTextCLIP is unreachable with the shipped config (SURVEY.md section 2 #2)")
    cfg = get_model_config(model_name)
    if cfg is None:
        raise RuntimeError(f"Model config for {model_name} not found.")
    if force_quick_gelu:
        cfg["quick_gelu"] = True
    openai = bool(pretrained) and pretrained.lower() == "openai"
    if openai:
        cfg["vision_cfg"]["image_size"] = img_size           # reference clip.py:112
    elif force_image_size is not None:
        cfg["vision_cfg"]["image_size"] = force_image_size   # reference clip.py:159-161 (img_size ignored here)
    model = CLIP(**cfg, cast_dtype=get_cast_dtype(precision), precision=precision)
    if pretrained:
        path = os.environ.get("AACLIP_CLIP_CKPT") or _MODEL_CKPT_PATHS.get(model_name)
        if not path or not Path(path).exists():
            raise RuntimeError(f"Model {model_name} not found; expected a local checkpoint at {path}")
        logging.info("Loading pretrained %s weights from %s", model_name, path)
        load_checkpoint(model, str(path), strict=True)
    elif require_pretrained:
        raise RuntimeError(
            f"Pretrained weights were required for (model: {model_name}, pretrained: {pretrained}) but not loaded.")
    model.to(device=device)
    model.visual.image_mean = (0.48145466, 0.4578275, 0.40821073)
    model.visual.image_std = (0.26862954, 0.26130258, 0.27577711)
    if output_dict and hasattr(model, "output_dict"):
        model.output_dict = True
    set_precision(model, precision)
    return model
