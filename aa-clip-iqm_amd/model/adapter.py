"""AdaptedCLIP on the MI355X HIP path: frozen CLIP towers + residual adapters +
4-level patch taps + seg/det projections (reference model/adapter.py:10-304).

Scope: the visual forward (adapter.py:137-184), the adapted
text encoder (:273-304) and the IQM side branch (:186-269, model/iqm.py): like the
reference, forward() runs the branch when text_embeddings is given and returns
None for iqm_outputs otherwise.  The reference never checkpoints the branch and
creates two of its layers with fresh random weights inside forward; here all of
its parameters are ordinary, seeded at construction and part of state_dict()
(model/iqm.py explains the two differences).
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from aaclip_hip import engine

from aaclip_hip._lib import ACT_LEAKY, ACT_RELU, EPI_ACT_F32, EPI_BIAS

from .adapter_modules import SimpleAdapter, SimpleProj
from .iqm import IQM, IQMOutput, sinusoidal_positions
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
        # ---- IQM side branch, reference adapter.py:56-92 (parameter names kept)
        h = iqm_hidden_size
        self.iqm_hidden_size = h
        self.iqm = IQM(hidden_size=h, num_hidden_layers=iqm_num_layers, num_attention_heads=iqm_num_heads,
                       encoder_hidden_size=h, text_encoder_hidden_size=768)
        self.class_query_mlp = nn.Sequential(nn.Linear(dv, h), nn.ReLU(), nn.Linear(h, h))
        self.query_adapters = nn.ModuleList([SimpleProj(dv, h, relu) for _ in range(len(levels))])
        self.visual_feature_proj = nn.Linear(h, h)       # the reference creates these two lazily, with random
        self.text_feature_proj = nn.Linear(2, 768)       # weights, inside forward (adapter.py:213-218,241-243)
        self.pos_embedding = nn.Parameter(sinusoidal_positions(512, h))
        self.visual_weight = nn.Parameter(torch.tensor(0.6))    # computed and unused by the reference too (:248-255)
        self.text_weight = nn.Parameter(torch.tensor(0.4))
        self.iqm_dropout = nn.Identity()
        self.iqm_layer_norm = nn.LayerNorm(h)
        for mod in (self.iqm, self.class_query_mlp, self.query_adapters):
            for p in mod.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)          # reference adapter.py:114-123
        self._zero_bias = {}
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
        adapters = self.image_adapter["layer_adapters"]
        n_levels = len(self.levels)
        blocks = list(v.transformer.resblocks)
        iqm_on = text_embeddings is not None
        # the IQM side branch has no split-fp16 GEMMs: under fp16x2 its products are exact fp32, EXCEPT the visual
        # cross-attention of the folded form below, which reads the fp16 halves of the tap rows as keys / values and
        # forms p.v with fp16 probabilities on the MFMA (csrc/iqm.hip cross_rows_mfma_kernel)
        icode = engine.plain_code(code)
        dt = engine.torch_dtype(icode)
        P = L - 1
        # 16-bit towers without the LeakyReLU in query_adapters: every step from the LayerNorm'ed tap rows to the keys and
        # values of the IQM cross-attention is linear, so the branch reads those rows as they are (_iqm_levels); otherwise
        # the levels are projected and concatenated like the reference does (_iqm_project_level)
        # (aaclip_cross_rows_levels takes at most 4 segments: more tap levels keep the projected form)
        fold_levels = (iqm_on and not self.relu and code in (engine.F16, engine.BF16, engine.F16X2)
                       and 2 * self.iqm.num_attention_heads <= 16 and xs.shape[-1] in (768, 1024)
                       and n_levels <= engine.CROSS_ROWS_MAX_SEGMENTS)
        ln_rows = []
        vis_cat = (torch.empty(B, n_levels * P, self.iqm_hidden_size, dtype=dt, device=xs.device)
                   if iqm_on and not fold_levels else None)
        # the whole tower with its adapters in ONE library call; the stream after every level stays in its own buffer
        # (no copies: the block behind a tap continues in a fresh one), the heads read them afterwards
        aws = [adapters[i].weight if i < self.image_adapt_until else None for i in range(len(blocks))]
        levels = [lv for lv in range(1, len(blocks) + 1) if lv in self.levels]
        xs, taps = v.transformer.run(xs, B, L, code, False, levels, adapter_weights=aws, mix=self.i_w)
        seg_tokens: List[torch.Tensor] = []
        det_token = None
        for k, tap in enumerate(taps):
            last = k == n_levels - 1
            res = engine.tap_head(
                tap, v.ln_post, self.image_adapter["seg_proj"][k].weight, self.relu, B, L, code,
                det_weight=self.image_adapter["det_proj"].weight if last else None, keep_rows=fold_levels)
            seg_tokens.append(res[0])
            if last:
                det_token = res[1]
            if fold_levels:
                ln_rows.append(res[2])
            elif iqm_on:
                self._iqm_project_level(tap, k, vis_cat, B, L, icode)
        if not iqm_on:
            return seg_tokens, det_token, None
        levels = self._iqm_levels(ln_rows, L, icode) if fold_levels else None
        return seg_tokens, det_token, self._iqm_branch(xs, vis_cat, text_embeddings, B, L, icode, levels)

    # -- the folded form of reference model/adapter.py:205-211: level k's rows stay ln_post(tap k) [B*L, D] (what the tap
    #    head computed anyway); query_adapters[k] enters the cross-attention as two concatenated weights
    def _iqm_levels(self, ln_rows, L, code):
        ws = [qa.weight for qa in self.query_adapters]
        key = (code,) + tuple((w.data_ptr(), w._version, str(w.device)) for w in ws)
        if getattr(self, "_qa_cat_key", None) != key:
            dt = engine.torch_dtype(code)
            with torch.no_grad():
                w_in = torch.cat([w.detach().float().t() for w in ws], 0).to(dt).contiguous()     # [levels*D, h]
                w_out = torch.cat([w.detach().float() for w in ws], 1).to(dt).contiguous()        # [h, levels*D]
            self._qa_cat, self._qa_cat_key = (w_in, w_out), key
        w_in, w_out = self._qa_cat
        return {"rows": ln_rows, "rows_per_image": L, "row0": 1, "keys": L - 1, "width": ws[0].shape[1],
                "w_in": w_in, "w_out": w_out}

    # -- reference model/adapter.py:205-208: query_adapters[k](ln_post(tap k)) for the patch rows, written into the
    #    slice of the concatenated visual features (torch.cat over dim 1, :210-211) that level k owns
    def _iqm_project_level(self, xs, k, vis_cat, B, L, code):
        v = self.image_encoder
        dt = engine.torch_dtype(code)
        h = self.iqm_hidden_size
        ln = engine.layernorm(xs, v.ln_post.weight, v.ln_post.bias, out_code=code)          # [B*L, D] compute dtype
        w = engine.CACHE.get(self.query_adapters[k].weight, code)
        if self.relu:
            tmp = torch.empty(B * L, h, dtype=torch.float32, device=xs.device)
            engine.gemm(code, EPI_ACT_F32, ln, w, None, tmp, act=ACT_LEAKY)
            tmp = tmp.to(dt)
        else:
            zb = self._zero_bias.get(xs.device)
            if zb is None or zb.numel() < h:
                zb = self._zero_bias[xs.device] = torch.zeros(h, dtype=torch.float32, device=xs.device)
            tmp = torch.empty(B * L, h, dtype=dt, device=xs.device)
            engine.gemm(code, EPI_BIAS, ln, w, zb, tmp)
        engine.drop_cls_rows(tmp, vis_cat, B, L, k * (L - 1), code)

    # -- reference model/adapter.py:186-269
    def _iqm_branch(self, xs, vis_cat, text_embeddings, B, L, code, levels=None):
        dt = engine.torch_dtype(code)
        h = self.iqm_hidden_size
        dev = xs.device
        te = text_embeddings.to(dev)
        if te.dim() != 3 or te.shape[0] != B or te.shape[-1] != 2:
            raise NotImplementedError(
                "the IQM branch is built for text_embeddings of shape [B, 768, 2] (what test_last.py:84 and train.py "
                "pass): the reference would re-create text_feature_proj with another in_features for any other form")
        # 1. queries: class_query_mlp(CLS row) for both, plus the first two sinusoidal positions (:191-203)
        cls = xs.view(B, L, -1)[:, 0, :].to(dt).contiguous()
        m0, m2 = self.class_query_mlp[0], self.class_query_mlp[2]
        t1 = torch.empty(B, h, dtype=torch.float32, device=dev)
        engine.gemm(code, EPI_ACT_F32, cls, engine.CACHE.get(m0.weight, code), engine._f32c(m0.bias), t1, act=ACT_RELU)
        cq = torch.empty(B, h, dtype=torch.float32, device=dev)
        engine.gemm(code, EPI_ACT_F32, t1.to(dt), engine.CACHE.get(m2.weight, code), engine._f32c(m2.bias), cq)
        pos = engine._f32c(self.pos_embedding)[:, :2, :].expand(B, 2, h).contiguous()
        query = engine.combine3(cq.unsqueeze(1).expand(B, 2, h).contiguous(), pos, None, 1.0, 1.0, 0.0)
        # 2. patch rows of all levels -> query space (:210-221): visual_feature_proj is NOT applied to the 5476 rows per
        #    image; it is folded into the two cross-attentions that read them (IQM._attend, enc_proj)
        vp = self.visual_feature_proj
        # 3. anchors [B, 768, 2] read as 768 tokens of width 2 -> Linear(2, 768) (:229-246)
        tp = self.text_feature_proj
        txt = engine.linear_smallk(te, tp.weight, tp.bias, code)
        out = self.iqm(query_embeds=query, query_length=2, encoder_hidden_states=vis_cat,
                       text_encoder_hidden_states=txt.view(B, te.shape[1], tp.weight.shape[0]), code=code,
                       encoder_proj=(vp.weight, vp.bias), encoder_levels=levels)
        hfin = engine.residual_layernorm(out.last_hidden_state.reshape(B * 2, h), None, self.iqm_layer_norm,
                                         self.iqm_layer_norm.eps)                             # :265-266
        return IQMOutput(hfin.view(B, 2, h), pooler_output=out.last_hidden_state.reshape(B, 2, h)[:, 0, :])

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
