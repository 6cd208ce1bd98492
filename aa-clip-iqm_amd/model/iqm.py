"""IQM side branch on the MI355X path: the "Improved Querying" transformer of reference
model/iqm.py (IQM, IQMEncoder, IQMLayer, IQM_Attention ...) and the glue around it in AdaptedCLIP.forward
(reference model/adapter.py:186-269), eval mode.

Two queries per image (normal / abnormal), started from an MLP of the CLS row plus a sinusoidal position, run through
`num_hidden_layers` layers of {self-attention, cross-attention to the projected patch rows of the four tap levels,
cross-attention to the projected anchors, fixed 0.4/0.3/0.3 fusion, GELU feed-forward}, then a LayerNorm.

Parameters carry the reference's names, so `state_dict()` of AdaptedCLIP matches the reference key for key
(`iqm.encoder.layer.0.crossattention.attention.key.weight`, `class_query_mlp.2.bias`, `query_adapters.1.fc.weight`,
`iqm_layer_norm.weight`, `pos_embedding` ...).  Two deliberate differences, both because the reference's behaviour
cannot be reproduced or checkpointed:
  * `visual_feature_proj` / `text_feature_proj` are created by the reference INSIDE forward with fresh random weights
    (adapter.py:213-218, 241-243) and never saved; here they are ordinary parameters (Linear(hidden, hidden) and
    Linear(2, 768): 2 because the anchors arrive as [B, 768, 2]) that are initialised once and saved with the rest;
  * dropout is the identity (eval); there is no training path.
The modules below are parameter containers: every product runs on the library's MFMA GEMM, the rest on the small
kernels of csrc/iqm.hip (aaclip_small_attention, aaclip_residual_layernorm, ...).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from aaclip_hip import engine
from aaclip_hip._lib import EPI_ACT_F32, EPI_BIAS, EPI_BIAS_GELU


class IQMOutput:
    """What the reference returns as iqm_outputs (a transformers BaseModelOutputWithPoolingAndCrossAttentions):
    callers read .last_hidden_state [B, 2, hidden] (test_last.py:104) and .pooler_output."""

    def __init__(self, last_hidden_state: torch.Tensor, pooler_output: Optional[torch.Tensor] = None):
        self.last_hidden_state = last_hidden_state
        # reference model/iqm.py:659-660: row 0 of the ENCODER output.  AdaptedCLIP later replaces last_hidden_state by
        # its LayerNorm (reference model/adapter.py:265) and leaves pooler_output as it was, i.e. pre-LayerNorm.
        self.pooler_output = last_hidden_state[:, 0, :] if pooler_output is None else pooler_output


class _SelfOutput(nn.Module):          # reference model/iqm.py:143-154 / :219-230
    def __init__(self, d_in: int, d: int, eps: float):
        super().__init__()
        self.dense = nn.Linear(d_in, d)
        self.LayerNorm = nn.LayerNorm(d, eps=eps)


class _MultiHeadAttention(nn.Module):  # reference model/iqm.py:23-58
    def __init__(self, d: int, d_kv: int):
        super().__init__()
        self.query = nn.Linear(d, d)
        self.key = nn.Linear(d_kv, d)
        self.value = nn.Linear(d_kv, d)


class _Attention(nn.Module):           # reference model/iqm.py:157-162
    def __init__(self, d: int, d_kv: int, eps: float):
        super().__init__()
        self.attention = _MultiHeadAttention(d, d_kv)
        self.output = _SelfOutput(d, d, eps)


class _Intermediate(nn.Module):        # reference model/iqm.py:206-217
    def __init__(self, d: int, inter: int):
        super().__init__()
        self.dense = nn.Linear(d, inter)


class IQMLayer(nn.Module):             # reference model/iqm.py:234-259
    def __init__(self, d: int, d_enc: int, d_txt: int, inter: int, eps: float):
        super().__init__()
        self.attention = _Attention(d, d, eps)
        self.crossattention = _Attention(d, d_enc, eps)
        self.text_crossattention = _Attention(d, d_txt, eps)
        self.intermediate = _Intermediate(d, inter)          # the non-query feed-forward: parameters only (the path
        self.output = _SelfOutput(inter, d, eps)             # never has more than query_length tokens, iqm.py:323)
        self.intermediate_query = _Intermediate(d, inter)
        self.output_query = _SelfOutput(inter, d, eps)


class _Encoder(nn.Module):
    def __init__(self, layers: int, d: int, d_enc: int, d_txt: int, inter: int, eps: float):
        super().__init__()
        self.layer = nn.ModuleList([IQMLayer(d, d_enc, d_txt, inter, eps) for _ in range(layers)])


class IQM(nn.Module):
    """reference model/iqm.py:497-673 (IQMConfig defaults :453-494: intermediate 2048, gelu, eps 1e-12,
    cross_attention_frequency 1)."""

    def __init__(self, hidden_size: int = 768, num_hidden_layers: int = 2, num_attention_heads: int = 8,
                 encoder_hidden_size: int = 768, text_encoder_hidden_size: int = 768, intermediate_size: int = 2048,
                 layer_norm_eps: float = 1e-12):
        super().__init__()
        if hidden_size % num_attention_heads:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (hidden_size, num_attention_heads))
        self.hidden_size, self.num_attention_heads, self.eps = hidden_size, num_attention_heads, layer_norm_eps
        self.layernorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.encoder = _Encoder(num_hidden_layers, hidden_size, encoder_hidden_size, text_encoder_hidden_size,
                                intermediate_size, layer_norm_eps)

    # ---- one IQM_Attention: q from h [B*nq, D] (fp32), k/v = Linear(enc) where enc is [B*Lk, Dk] in the compute dtype
    def _attend(self, att: _Attention, h: torch.Tensor, enc: Optional[torch.Tensor], B: int, nq: int, Lk: int,
                code: int, enc_proj=None, enc_levels=None) -> torch.Tensor:
        dt = engine.torch_dtype(code)
        D = self.hidden_size
        hq = h.to(dt)                                      # [B*nq, D]: 2 rows per image
        q = torch.empty(B * nq, D, dtype=torch.float32, device=h.device)
        engine.gemm(code, EPI_ACT_F32, hq, engine.CACHE.get(att.attention.query.weight, code),
                    engine._f32c(att.attention.query.bias), q)
        H = self.num_attention_heads
        if enc_levels is not None:
            # the rows are level s's LayerNorm'ed tap rows t, enc = P (W_qa[s] t) + b_p (query_adapters, torch.cat,
            # visual_feature_proj: reference model/adapter.py:205-221), every step linear: all of it moves to the query
            # side and behind the weighted sums (include/aaclip.h, aaclip_cross_rows_levels) -- no per-row product at all
            lv = enc_levels
            pw, pb = enc_proj
            qm = engine.head_expand(q, H, 1.0 / math.sqrt(D // H), code)                         # [B*nq*H, D]
            kin = att.attention.key.weight.shape[1]
            qt = torch.empty(B * nq * H, kin, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, qm, engine.CACHE.get(att.attention.key.weight, code, "transpose"), None, qt)
            qx = torch.empty(B * nq * H, pw.shape[1], dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, qt.to(dt), engine.CACHE.get(pw, code, "transpose"), None, qx)
            nseg, Dk = len(lv["rows"]), lv["width"]
            u = torch.empty(B * nq * H, nseg * Dk, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, qx.to(dt), lv["w_in"], None, u)        # u[., s] = W_qa[s]^T qx
            tbar = engine.cross_rows_levels(u, lv["rows"], B, nq * H, lv["rows_per_image"], lv["row0"], lv["keys"], Dk)
            xbar = torch.empty(B * nq * H, pw.shape[1], dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, tbar.to(dt), lv["w_out"], None, xbar)  # sum_s W_qa[s] tbar[., s]
            ebar = torch.empty(B * nq * H, kin, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, xbar.to(dt), engine.CACHE.get(pw, code), engine._f32c(pb), ebar)
            full = torch.empty(B * nq * H, D, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, ebar.to(dt), engine.CACHE.get(att.attention.value.weight, code),
                        engine._f32c(att.attention.value.bias), full)
            ctx = engine.head_diag(full, H)
            dense = torch.empty(B * nq, D, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, ctx.to(dt), engine.CACHE.get(att.output.dense.weight, code),
                        engine._f32c(att.output.dense.bias), dense)
            return engine.residual_layernorm(dense, h, att.output.LayerNorm, self.eps)
        if enc is not None and (nq * H) % 4 == 0 and nq * H <= 16 and enc.shape[-1] in (256, 512, 768, 1024):
            # cross-attention over MANY rows for a handful of queries: W_k moves to the query side and W_v behind the
            # probability-weighted sum of the raw rows (include/aaclip.h, aaclip_cross_rows): the reference's key /
            # value projections of all Lk rows (2 x Lk x Dk x D MACs per image, reference model/iqm.py:116-121) become
            # two [nq*H, .] products.  b_k only shifts every score of a row by the same amount: softmax-invariant.
            qm = engine.head_expand(q, H, 1.0 / math.sqrt(D // H), code)                         # [B*nq*H, D]
            kin = att.attention.key.weight.shape[1]
            qt = torch.empty(B * nq * H, kin, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, qm, engine.CACHE.get(att.attention.key.weight, code, "transpose"), None, qt)
            if enc_proj is not None:
                # the rows are enc = P x + b_p of raw rows x (AdaptedCLIP.visual_feature_proj on the concatenated levels,
                # reference model/adapter.py:213-221): the same algebra once more -- (P^T qt) . x_j + const on the way
                # in, P (sum_j p_j x_j) + b_p on the way out -- and the [B*Lk, .] projection is never computed
                pw, pb = enc_proj
                qx = torch.empty(B * nq * H, pw.shape[1], dtype=torch.float32, device=h.device)
                engine.gemm(code, EPI_ACT_F32, qt.to(dt), engine.CACHE.get(pw, code, "transpose"), None, qx)
                xbar = engine.cross_rows(qx, enc, B, nq * H, Lk, code)
                ebar = torch.empty(B * nq * H, kin, dtype=torch.float32, device=h.device)
                engine.gemm(code, EPI_ACT_F32, xbar.to(dt), engine.CACHE.get(pw, code), engine._f32c(pb), ebar)
            elif enc.dtype in (torch.float16, torch.bfloat16) and enc.shape[-1] in (768, 1024):
                # 16-bit rows: the matrix-core kernel, one segment (the anchor tokens of the text cross-attention)
                ebar = engine.cross_rows_levels(qt, [enc], B, nq * H, Lk, 0, Lk, enc.shape[-1])
            else:
                ebar = engine.cross_rows(qt, enc, B, nq * H, Lk, code)                           # [B*nq*H, Dk] fp32
            full = torch.empty(B * nq * H, D, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, ebar.to(dt), engine.CACHE.get(att.attention.value.weight, code),
                        engine._f32c(att.attention.value.bias), full)
            ctx = engine.head_diag(full, H)
            dense = torch.empty(B * nq, D, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, ctx.to(dt), engine.CACHE.get(att.output.dense.weight, code),
                        engine._f32c(att.output.dense.bias), dense)
            return engine.residual_layernorm(dense, h, att.output.LayerNorm, self.eps)
        src = hq if enc is None else enc
        k = torch.empty(src.shape[0], D, dtype=dt, device=h.device)      # compute dtype (fp32 on the fp32 path)
        v = torch.empty_like(k)
        engine.gemm(code, EPI_BIAS, src, engine.CACHE.get(att.attention.key.weight, code),
                    engine._f32c(att.attention.key.bias), k)
        engine.gemm(code, EPI_BIAS, src, engine.CACHE.get(att.attention.value.weight, code),
                    engine._f32c(att.attention.value.bias), v)
        ctx = engine.small_attention(q, k, v, B, nq, Lk, self.num_attention_heads, code)
        dense = torch.empty(B * nq, D, dtype=torch.float32, device=h.device)
        engine.gemm(code, EPI_ACT_F32, ctx.to(dt), engine.CACHE.get(att.output.dense.weight, code),
                    engine._f32c(att.output.dense.bias), dense)
        return engine.residual_layernorm(dense, h, att.output.LayerNorm, self.eps)

    def forward(self, query_embeds: torch.Tensor, query_length: Optional[int] = None,
                encoder_hidden_states: Optional[torch.Tensor] = None,
                text_encoder_hidden_states: Optional[torch.Tensor] = None, code: Optional[int] = None,
                encoder_proj=None, encoder_levels=None, **_unused):
        """query_embeds fp32 [B, nq, D]; encoder_hidden_states [B, Lv, D] and text_encoder_hidden_states [B, Lt, D] in
        the compute dtype (or fp32) -> IQMOutput.  reference model/iqm.py:572-673 with all masks zero.
        encoder_proj = (weight, bias): encoder_hidden_states are the rows BEFORE that Linear; it is folded into the
        cross-attention (see _attend) instead of being applied to every row.
        encoder_levels (instead of encoder_hidden_states; AdaptedCLIP._iqm_levels builds it): the LayerNorm'ed rows of
        the tap levels themselves plus the concatenated query_adapters weights -- the level projections fold too."""
        engine.require_gpu(query_embeds, "IQM")
        if code is None:
            code = engine.dtype_code(getattr(self, "precision", "fp32"))
        # no split-fp16 GEMMs on this side branch: fp32 products under fp16x2 (with encoder_levels the visual
        # cross-attention still reads fp16 key / value rows and fp16 probabilities, see AdaptedCLIP.forward)
        code = engine.plain_code(code)
        dt = engine.torch_dtype(code)
        B, nq, D = query_embeds.shape
        if (encoder_hidden_states is None and encoder_levels is None) or text_encoder_hidden_states is None:
            raise ValueError("encoder_hidden_states must be given for cross-attention layers")     # iqm.py:289
        if encoder_levels is not None:
            if encoder_proj is None or (nq * self.num_attention_heads) > 16:
                raise ValueError("encoder_levels needs encoder_proj and at most 16 queries x heads per image")
            vis, Lv = None, encoder_levels["keys"] * len(encoder_levels["rows"])
        else:
            vis = encoder_hidden_states.to(dt).reshape(-1, encoder_hidden_states.shape[-1]).contiguous()
            Lv = encoder_hidden_states.shape[1]
        txt = text_encoder_hidden_states.to(dt).reshape(-1, text_encoder_hidden_states.shape[-1]).contiguous()
        Lt = text_encoder_hidden_states.shape[1]
        h = engine.residual_layernorm(engine._f32c(query_embeds).reshape(B * nq, D), None, self.layernorm, self.eps)
        for layer in self.encoder.layer:
            a = self._attend(layer.attention, h, None, B, nq, nq, code)
            c = self._attend(layer.crossattention, a, vis, B, nq, Lv, code, enc_proj=encoder_proj,
                             enc_levels=encoder_levels)
            t = self._attend(layer.text_crossattention, c, txt, B, nq, Lt, code)
            mix = engine.combine3(a, c, t, 0.4, 0.3, 0.3)                                            # iqm.py:311-315
            inter = torch.empty(B * nq, layer.intermediate_query.dense.weight.shape[0], dtype=dt, device=h.device)
            engine.gemm(code, EPI_BIAS_GELU, mix.to(dt), engine.CACHE.get(layer.intermediate_query.dense.weight, code),
                        engine._f32c(layer.intermediate_query.dense.bias), inter)
            dense = torch.empty(B * nq, D, dtype=torch.float32, device=h.device)
            engine.gemm(code, EPI_ACT_F32, inter, engine.CACHE.get(layer.output_query.dense.weight, code),
                        engine._f32c(layer.output_query.dense.bias), dense)
            h = engine.residual_layernorm(dense, mix, layer.output_query.LayerNorm, self.eps)
        return IQMOutput(h.view(B, nq, D))


def sinusoidal_positions(max_len: int, d_model: int) -> torch.Tensor:
    """reference model/adapter.py:98-105 -> [1, max_len, d_model]."""
    position = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)
