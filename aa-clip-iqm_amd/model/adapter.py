"""AdaptedCLIP on the MI355X HIP path: frozen CLIP towers + residual adapters +
4-level patch taps + seg/det projections (reference model/adapter.py:10-304).

Scope: the visual forward (adapter.py:137-184) and the
adapted text encoder (:273-304) are built.  The IQM side branch (:186-269) is
not: its weights are never checkpointed by the reference and it is outside the
north-star path, so forward() always returns None for iqm_outputs, exactly what
the reference returns when text_embeddings is None.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from aaclip_hip import engine

from .adapter_modules import SimpleAdapter, SimpleProj
from .transformer import set_precision


class AdaptedCLIP(nn.Module):
    def __init__(self, clip_model, text_adapt_weight: float = 0.1, image_adapt_weight: float = 0.1,
                 text_adapt_until: int = 3, image_adapt_until: int = 6, levels: list = [6, 12, 18, 24],
                 relu: bool = True, iqm_hidden_size: int = 768, iqm_num_layers: int = 2, iqm_num_heads: int = 8,
                 **kwargs):
        super().__init__()
        self.clipmodel = clip_model
        self.image_encoder = clip_model.visual
        self.text_adapt_until = text_adapt_until
        self.image_adapt_until = image_adapt_until
        self.t_w = text_adapt_weight
        self.i_w = image_adapt_weight
        self.levels = list(levels)
        self.relu = relu
        dv = self.image_encoder.embed_dim
        dt = clip_model.transformer.width
        e = self.image_encoder.output_dim
        self.image_adapter = nn.ModuleDict({
            "layer_adapters": nn.ModuleList([SimpleAdapter(dv, dv) for _ in range(image_adapt_until)]),
            "seg_proj": nn.ModuleList([SimpleProj(dv, e, relu) for _ in range(len(levels))]),
            "det_proj": SimpleProj(dv, e, relu),
        })
        self.text_adapter = nn.ModuleList(
            [SimpleAdapter(dt, dt) for _ in range(text_adapt_until)] + [SimpleProj(dt, e, relu=True)])
        for mod in (self.image_adapter, self.text_adapter):
            for p in mod.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)          # reference adapter.py:107-113
        self.iqm = None                                  # out of scope, see module docstring
        set_precision(self, getattr(clip_model, "precision", "fp32"))

    def _code(self) -> int:
        return engine.dtype_code(getattr(self.clipmodel, "precision", "fp32"))

    # -- reference model/adapter.py:125-135
    def forward_original(self, x, modality="visual"):
        if modality != "visual":
            raise ValueError("modality must be visual")
        cls_features, patch_features = self.clipmodel.encode_image(x, [self.image_encoder.transformer.layers])
        v = self.clipmodel.visual
        patch_features = [v._global_pool(t)[1] for t in patch_features]
        patch_features = [v.ln_post(t) for t in patch_features]
        code = self._code()
        patch_features = [engine.linear(t, v.proj, None, False, code, "transpose") for t in patch_features]
        return patch_features, cls_features

    # -- reference model/adapter.py:137-184
    def forward(self, x, text_embeddings=None, iqm_hidden_size=None):
        code = self._code()
        v = self.image_encoder
        xs, B, L = engine.patch_embed(x, v, code)
        heads = v.num_heads
        adapters = self.image_adapter["layer_adapters"]
        seg_tokens: List[torch.Tensor] = []
        det_token = None
        n_levels = len(self.levels)
        blocks = list(v.transformer.resblocks)
        run, run_aw = [], []   # consecutive blocks up to the next tap: one aaclip_blocks call
        for i, blk in enumerate(blocks):
            run.append(blk)
            run_aw.append(adapters[i].weight if i < self.image_adapt_until else None)
            if (i + 1) in self.levels or i + 1 == len(blocks):
                engine.run_blocks(xs, run, B, L, heads, code, causal=False, adapter_weights=run_aw, mix=self.i_w)
                run, run_aw = [], []
            if (i + 1) in self.levels:
                k = len(seg_tokens)
                last = k == n_levels - 1
                seg, det = engine.tap_head(
                    xs, v.ln_post, self.image_adapter["seg_proj"][k].weight, self.relu, B, L, code,
                    det_weight=self.image_adapter["det_proj"].weight if last else None)
                seg_tokens.append(seg)
                if last:
                    det_token = det
        return seg_tokens, det_token, None

    # -- reference model/adapter.py:273-304
    def encode_text(self, text, adapt_text=True):
        if not adapt_text:
            return self.clipmodel.encode_text(text)
        code = self._code()
        c = self.clipmodel
        x, tk = engine.text_embed(text, c.token_embedding.weight, c.positional_embedding)
        n, T = tk.shape
        heads = c.transformer.heads
        blocks = list(c.transformer.resblocks)
        aws = [self.text_adapter[i].weight if i < self.text_adapt_until else None for i in range(len(blocks))]
        engine.run_blocks(x, blocks, n, T, heads, code, causal=True, adapter_weights=aws, mix=self.t_w)
        return engine.row_head(x, tk, c.ln_final, self.text_adapter[-1].weight, "plain", True, n, T, 0, code)
