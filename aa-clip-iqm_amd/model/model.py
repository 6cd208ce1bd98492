"""CLIP container with the reference's attribute names and state-dict keys
(reference model/model.py:149-201), executed by the HIP path."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple, Union

import numpy as np
import torch
from torch import nn

from aaclip_hip import engine

from .transformer import LayerNorm, Transformer, VisionTransformer, causal_attn_mask, set_precision, to_2tuple


@dataclass
class CLIPVisionCfg:
    layers: int = 12
    width: int = 768
    head_width: int = 64
    mlp_ratio: float = 4.0
    patch_size: int = 16
    image_size: Union[Tuple[int, int], int] = 224
    patch_dropout: float = 0.0


@dataclass
class CLIPTextCfg:
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 512
    heads: int = 8
    layers: int = 12


def get_cast_dtype(precision: str):
    """reference model/model.py:63-69."""
    return {"bf16": torch.bfloat16, "fp16": torch.float16}.get(precision)


class CLIP(nn.Module):
    def __init__(self, embed_dim: int, vision_cfg, text_cfg, quick_gelu: bool = False, cast_dtype=None,
                 output_dict: bool = False, precision: str = "fp32"):
        super().__init__()
        if quick_gelu:
            raise NotImplementedError("the reference hot path uses exact-erf nn.GELU (model/model.py:84); "
                                      "QuickGELU is not built")
        if isinstance(vision_cfg, dict):
            vision_cfg = CLIPVisionCfg(**{k: v for k, v in vision_cfg.items() if k in CLIPVisionCfg.__dataclass_fields__})
        if isinstance(text_cfg, dict):
            text_cfg = CLIPTextCfg(**{k: v for k, v in text_cfg.items() if k in CLIPTextCfg.__dataclass_fields__})
        if isinstance(vision_cfg.layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet towers are outside the AA-CLIP hot path (SURVEY.md section 2 #9)")
        self.output_dict = output_dict
        heads = vision_cfg.width // vision_cfg.head_width
        self.visual = VisionTransformer(
            image_size=vision_cfg.image_size, patch_size=vision_cfg.patch_size, width=vision_cfg.width,
            layers=vision_cfg.layers, heads=heads, mlp_ratio=vision_cfg.mlp_ratio, output_dim=embed_dim)
        self.transformer = Transformer(text_cfg.width, text_cfg.layers, text_cfg.heads)
        self.vocab_size = text_cfg.vocab_size
        self.context_length = text_cfg.context_length
        self.token_embedding = nn.Embedding(text_cfg.vocab_size, text_cfg.width)
        self.positional_embedding = nn.Parameter(torch.empty(text_cfg.context_length, text_cfg.width))
        self.ln_final = LayerNorm(text_cfg.width)
        self.text_projection = nn.Parameter(torch.empty(text_cfg.width, embed_dim))
        self.register_buffer("attn_mask", causal_attn_mask(text_cfg.context_length), persistent=False)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=text_cfg.width ** -0.5)
        set_precision(self, precision)

    # -- reference model/model.py:185-188
    def encode_image(self, image, out_layers, normalize: bool = False):
        pooled, tokens = self.visual(image, out_layers)
        if normalize:
            raise NotImplementedError("normalize=True is not used on the AA-CLIP path")
        return pooled, tokens

    # -- reference model/model.py:190-201
    def encode_text(self, text, normalize: bool = False):
        code = engine.dtype_code(getattr(self, "precision", "fp32"))
        x, tk = engine.text_embed(text, self.token_embedding.weight, self.positional_embedding)
        n, T = tk.shape
        x, _ = self.transformer.run(x, n, T, code, True)
        out = engine.row_head(x, tk, self.ln_final, self.text_projection, "transpose", False, n, T, 0, code)
        if normalize:
            raise NotImplementedError("normalize=True is not used on the AA-CLIP path")
        return out


def resize_pos_embed(state_dict, model, interpolation: str = "bicubic", antialias: bool = True):
    """Load-time host step, reference model/model.py:396-427: bicubic+antialias
    resize of the positional-embedding grid to the model's grid, CLS row kept."""
    old = state_dict.get("visual.positional_embedding", None)
    if old is None or not hasattr(model.visual, "grid_size"):
        return
    grid = to_2tuple(model.visual.grid_size)
    new_len = grid[0] * grid[1] + 1
    if new_len == old.shape[0]:
        return
    tok, img = old[:1], old[1:]
    g = int(math.sqrt(len(img)))
    img = img.reshape(1, g, g, -1).permute(0, 3, 1, 2)
    img = torch.nn.functional.interpolate(img.float(), size=grid, mode=interpolation, antialias=antialias,
                                          align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(grid[0] * grid[1], -1).to(old.dtype)
    state_dict["visual.positional_embedding"] = torch.cat([tok, img], dim=0)
