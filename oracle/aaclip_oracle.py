"""CPU oracle for the AA-CLIP hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A straight-line, batch-first restatement (stock torch CPU ops, fp32 by default,
fp64 on request) of the reference algorithm that SURVEY.md section 8(a) scopes:
patch embed -> cls/pos/ln_pre -> 24 pre-LN residual attention blocks with the
residual adapters mixed in -> 4 tap levels -> ln_post/seg_proj/normalise ->
patch x text-anchor similarity map, plus the text tower and the anchor builder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The product package never imports it.

Pinning: every function here is checked in tests/test_oracle_golden.py against
golden vectors produced by running the reference itself (imported from
/root/reference inside the build container by tests/golden/make_golden.py; the
vectors are committed under tests/golden/).  One piece is NOT pinned by the
reference: the test-mode Gaussian blur is kornia==0.6.9's gaussian_blur2d
(reference requirements.txt:3, call site forward_utils.py:208-210); kornia is
not installed here, so `gaussian_blur2d` below restates its published algorithm
(separable normalised taps exp(-x^2/2s^2), reflect border) -- "parity unpinned"
for that one function; everything around it is pinned.

Each function cites the reference file:line it follows.  Weights are passed as a
plain dict of tensors with the reference's state-dict key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """reference model/transformer.py:37-43 (nn.LayerNorm, eps 1e-5)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() exact-erf form (reference model/model.py:84: QuickGELU is NOT used)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def leaky_relu(x: torch.Tensor, slope: float = 0.01) -> torch.Tensor:
    """nn.LeakyReLU() default slope (reference model/adapter_modules.py:9)."""
    return torch.where(x >= 0, x, x * slope)


def causal_mask(n: int, dtype: torch.dtype) -> torch.Tensor:
    """reference model/transformer.py:629-635: -inf strictly above the diagonal."""
    m = torch.full((n, n), float("-inf"), dtype=dtype)
    return torch.triu(m, diagonal=1)


def self_attention(h: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor,
                   out_w: torch.Tensor, out_b: torch.Tensor, heads: int,
                   mask: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.MultiheadAttention forward (math path) as the reference calls it at
    model/transformer.py:200,237: packed in_proj, q scaled by head_dim^-1/2,
    additive mask, softmax over keys, out_proj.  h is [B, L, D]."""
    B, L, D = h.shape
    hd = D // heads
    qkv = h @ in_w.t() + in_b
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.view(B, L, heads, hd).transpose(1, 2)
    v = v.view(B, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B, L, D)
    return ctx @ out_w.t() + out_b


def vv_attention_batch_axis(h: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor,
                            out_w: torch.Tensor, out_b: torch.Tensor, heads: int) -> torch.Tensor:
    """The "surgery" Attention of reference model/transformer.py:102-152 as it actually runs after
    DAPM_replace (:406-425): q = k = v (the value projection), scores v.v^T * head_dim^-1/2, softmax,
    out_proj.  The block feeds it the LND stream but the module unpacks `B, N, C = q_x.shape`
    (:126), so the softmax runs over the BATCH axis, separately for every token position and head
    (outputs depend on the batch composition).  h is batch-first [B, L, D]."""
    B, L, D = h.shape
    hd = D // heads
    v = (h @ in_w.t() + in_b)[..., 2 * D:].view(B, L, heads, hd).permute(1, 2, 0, 3)   # [L, H, B, hd]
    p = torch.softmax((v @ v.transpose(-1, -2)) * (hd ** -0.5), dim=-1)                # [L, H, B, B]
    ctx = (p @ v).permute(2, 0, 1, 3).reshape(B, L, D)
    return ctx @ out_w.t() + out_b


def resblock(x: torch.Tensor, sd: SD, p: str, heads: int, mask: Optional[torch.Tensor],
             surgery: bool = False) -> torch.Tensor:
    """ResidualAttentionBlock.forward, reference model/transformer.py:239-258
    (ls_1/ls_2 are Identity, no ln_1_kv).  surgery=True: the block's attention was swapped by
    DAPM_replace for the V-V module above (same weights)."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    if surgery:
        x = x + vv_attention_batch_axis(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                                        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads)
    else:
        x = x + self_attention(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                               sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, mask)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = gelu_erf(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
    return x + (h @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"])


def adapter_mix(x: torch.Tensor, w: torch.Tensor, weight: float) -> torch.Tensor:
    """reference model/adapter.py:163-170 (and :288-295 for text):
    a = LeakyReLU(x W^T); a = a * |x| / |a| per token (no eps); x = w*a + (1-w)*x."""
    a = leaky_relu(x @ w.t())
    a = a * x.norm(dim=-1, keepdim=True) / a.norm(dim=-1, keepdim=True)
    return weight * a + (1 - weight) * x


# --------------------------------------------------------------------------
# visual side
# --------------------------------------------------------------------------
def patch_embed(img: torch.Tensor, conv_w: torch.Tensor) -> torch.Tensor:
    """reference model/transformer.py:359-365,507-509 / model/adapter.py:139-141:
    Conv2d(3->D, k=s=patch, no bias) == GEMM over unfolded patches; -> [B, P, D]."""
    B, C, H, W = img.shape
    D, _, ps, _ = conv_w.shape
    g = H // ps
    cols = img[:, :, : g * ps, : g * ps].reshape(B, C, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5)
    cols = cols.reshape(B, g * g, C * ps * ps)
    return cols @ conv_w.reshape(D, -1).t()


def visual_stem(img: torch.Tensor, sd: SD) -> torch.Tensor:
    """reference model/adapter.py:139-158 (== model/transformer.py:507-528):
    patches, prepend class embedding, add positional embedding, ln_pre."""
    x = patch_embed(img, sd["visual.conv1.weight"])
    cls = sd["visual.class_embedding"].to(x.dtype).expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].to(x.dtype)
    return layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])


def _cast(sd: SD, dtype: torch.dtype) -> SD:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def encode_image(img: torch.Tensor, sd: SD, heads: int, out_layers: Sequence[int],
                 dtype: torch.dtype = torch.float32, dpam_layer: Optional[int] = None
                 ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """CLIP.encode_image(image, out_layers), reference model/model.py:185-188 ->
    model/transformer.py:490-551 -> :295-317.  Taps are the raw residual stream
    (incl. CLS) after 1-based layer idx in out_layers; pooled = ln_post(x[:,0]) @ proj.
    dpam_layer: as after visual.DAPM_replace(DPAM_layer) (:406-425) -- the LAST dpam_layer-1
    blocks use the V-V attention."""
    sd = _cast(sd, dtype)
    x = visual_stem(img.to(dtype), sd)
    layers = sum(1 for k in sd if k.startswith("visual.transformer.resblocks.") and k.endswith("ln_1.weight"))
    n_surgery = max(dpam_layer - 1, 0) if dpam_layer is not None else 0
    taps = []
    for i in range(layers):
        x = resblock(x, sd, f"visual.transformer.resblocks.{i}.", heads, None, surgery=i >= layers - n_surgery)
        if (i + 1) in out_layers:
            taps.append(x)
    pooled = layer_norm(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
    return pooled, taps


def adapted_visual_forward(img: torch.Tensor, sd: SD, ia: SD, heads: int,
                           image_adapt_weight: float = 0.1, image_adapt_until: int = 6,
                           levels: Sequence[int] = (6, 12, 18, 24), relu: bool = False,
                           dtype: torch.dtype = torch.float32,
                           return_stream: bool = False):
    """AdaptedCLIP.forward(x) with text_embeddings=None, reference
    model/adapter.py:137-184.  `ia` is the image_adapter state dict.
    Returns (seg_tokens: list of [B,P,E] unit rows, det_token [B,E])."""
    sd, ia = _cast(sd, dtype), _cast(ia, dtype)
    pk = "fc.0.weight" if relu else "fc.weight"
    x = visual_stem(img.to(dtype), sd)
    layers = sum(1 for k in sd if k.startswith("visual.transformer.resblocks.") and k.endswith("ln_1.weight"))
    taps, stream = [], []
    for i in range(layers):
        x = resblock(x, sd, f"visual.transformer.resblocks.{i}.", heads, None)
        if i < image_adapt_until:
            x = adapter_mix(x, ia[f"layer_adapters.{i}.fc.0.weight"], image_adapt_weight)
        if (i + 1) in levels:
            taps.append(x[:, 1:, :])
            stream.append(x)
    taps = [layer_norm(t, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) for t in taps]
    seg = []
    for i, t in enumerate(taps):
        s = t @ ia[f"seg_proj.{i}.{pk}"].t()
        if relu:
            s = leaky_relu(s)
        seg.append(F.normalize(s, dim=-1))
    d = taps[-1] @ ia[f"det_proj.{pk}"].t()
    if relu:
        d = leaky_relu(d)
    det = F.normalize(d, dim=-1).mean(dim=1)
    if return_stream:
        return seg, det, stream
    return seg, det


# --------------------------------------------------------------------------
# IQM side branch
# --------------------------------------------------------------------------
def _iqm_linear(x: torch.Tensor, sd: SD, name: str) -> torch.Tensor:
    y = x @ sd[name + ".weight"].t()
    return y + sd[name + ".bias"] if (name + ".bias") in sd else y


def _iqm_attention(h: torch.Tensor, enc: Optional[torch.Tensor], sd: SD, p: str, heads: int) -> torch.Tensor:
    """IQM_Attention = IQM_MultiHeadAttention + IQM_SelfOutput, reference model/iqm.py:23-139,143-154,157-204:
    q from the hidden states, k/v from `enc` (cross) or from the hidden states (self); scores / sqrt(head size)
    (:116), masks are all-zero on this path (:621-642), dropout is the identity in eval; then
    LayerNorm(dense(context) + hidden_states) with eps 1e-12."""
    B, nq, D = h.shape
    src = h if enc is None else enc
    hd = D // heads
    q = _iqm_linear(h, sd, p + "attention.query").view(B, nq, heads, hd).transpose(1, 2)
    k = _iqm_linear(src, sd, p + "attention.key").view(B, src.shape[1], heads, hd).transpose(1, 2)
    v = _iqm_linear(src, sd, p + "attention.value").view(B, src.shape[1], heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    ctx = (a @ v).transpose(1, 2).reshape(B, nq, D)
    return layer_norm(_iqm_linear(ctx, sd, p + "output.dense") + h, sd[p + "output.LayerNorm.weight"],
                      sd[p + "output.LayerNorm.bias"], 1e-12)


def iqm_branch(x_final: torch.Tensor, tokens_ln: Sequence[torch.Tensor], text_embeddings: torch.Tensor, isd: SD,
               relu: bool = False, heads: int = 8, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """The IQM glue of AdaptedCLIP.forward (reference model/adapter.py:186-269) + IQM.forward (model/iqm.py:572-673)
    in eval mode -> iqm_outputs.last_hidden_state [B, 2, hidden].
      x_final    [B, L, 1024]   the residual stream after the last block (CLS row used, adapter.py:191)
      tokens_ln  4 x [B, P, 1024]  ln_post of the tapped patch rows (adapter.py:174)
      text_embeddings [B, 768, 2]  as test_last.py:84 passes them: the branch reads this as 768 "tokens" of width 2
                                   (adapter.py:229-243), reproduced as executed
    `isd`: aaclip_hip.synth.synth_iqm_state_dict keys (the reference's parameter names)."""
    isd = _cast(isd, dtype)
    x_final, text_embeddings = x_final.to(dtype), text_embeddings.to(dtype)
    B = x_final.shape[0]
    cq = torch.relu(_iqm_linear(x_final[:, 0, :], isd, "class_query_mlp.0"))
    cq = _iqm_linear(cq, isd, "class_query_mlp.2").unsqueeze(1).repeat(1, 2, 1)           # adapter.py:192-196
    query = cq + isd["pos_embedding"][:, :2, :]                                             # :199-203
    pk = "fc.0" if relu else "fc"
    vis = []
    for i, t in enumerate(tokens_ln):                                                       # :206-208
        v = t.to(dtype) @ isd[f"query_adapters.{i}.{pk}.weight"].t()
        vis.append(leaky_relu(v) if relu else v)
    vis = _iqm_linear(torch.cat(vis, dim=1), isd, "visual_feature_proj")                    # :210-221 (cat over dim 1)
    assert text_embeddings.dim() == 3 and text_embeddings.shape[1] != 2                     # the branch of :229-235 taken
    txt = _iqm_linear(text_embeddings, isd, "text_feature_proj")                            # :241-246  [B, 768, 768]
    h = layer_norm(query, isd["iqm.layernorm.weight"], isd["iqm.layernorm.bias"], 1e-12)    # iqm.py:617
    layers = sum(1 for k in isd if k.startswith("iqm.encoder.layer.") and k.endswith("attention.attention.query.weight")
                 and ".crossattention." not in k and ".text_crossattention." not in k)
    for l in range(layers):                                                                 # IQMLayer.forward :261-343
        p = f"iqm.encoder.layer.{l}."
        a = _iqm_attention(h, None, isd, p + "attention.", heads)
        c = _iqm_attention(a, vis, isd, p + "crossattention.", heads)
        t = _iqm_attention(c, txt, isd, p + "text_crossattention.", heads)
        mix = 0.4 * a + 0.3 * c + 0.3 * t                                                   # :311-315
        inter = gelu_erf(_iqm_linear(mix, isd, p + "intermediate_query.dense"))             # :350-353 (hidden_act gelu)
        h = layer_norm(_iqm_linear(inter, isd, p + "output_query.dense") + mix, isd[p + "output_query.LayerNorm.weight"],
                       isd[p + "output_query.LayerNorm.bias"], 1e-12)
    return layer_norm(h, isd["iqm_layer_norm.weight"], isd["iqm_layer_norm.bias"], 1e-5)   # adapter.py:265


def iqm_anomaly_map(seg_tokens: Sequence[torch.Tensor], h: torch.Tensor, img_size: int) -> torch.Tensor:
    """reference test_last.py:102-138,144: per level sigmoid(cos(f, q_abnormal) - cos(f, q_normal)) on the patch grid,
    bilinear (align_corners=False) to img_size, summed over levels -> [B, S, S].  (The reference only projects the
    queries when their width differs from the patch features'; both are 768 here.)"""
    nq, aq = h[:, 0, :], h[:, 1, :]
    total = 0
    for f in seg_tokens:
        B, P, _ = f.shape
        g = int(round(P ** 0.5))
        pred = torch.sigmoid(F.cosine_similarity(f, aq.unsqueeze(1), dim=-1) - F.cosine_similarity(f, nq.unsqueeze(1), dim=-1))
        total = total + F.interpolate(pred.view(B, 1, g, g), size=(img_size, img_size), mode="bilinear",
                                      align_corners=False)[:, 0]
    return total


def adapted_visual_forward_iqm(img: torch.Tensor, sd: SD, ia: SD, isd: SD, text_embeddings: torch.Tensor, heads: int,
                               relu: bool = False, dtype: torch.dtype = torch.float32):
    """AdaptedCLIP.forward(x, text_embeddings) (reference model/adapter.py:137-271) -> (seg, det, last_hidden_state)."""
    seg, det, stream = adapted_visual_forward(img, sd, ia, heads, relu=relu, dtype=dtype, return_stream=True)
    sdc = _cast(sd, dtype)
    tokens_ln = [layer_norm(x[:, 1:, :], sdc["visual.ln_post.weight"], sdc["visual.ln_post.bias"]) for x in stream]
    return seg, det, iqm_branch(stream[-1], tokens_ln, text_embeddings, isd, relu=relu, dtype=dtype)


# --------------------------------------------------------------------------
# text side
# --------------------------------------------------------------------------
def _text_layers(sd: SD) -> int:
    return sum(1 for k in sd if k.startswith("transformer.resblocks.") and k.endswith("ln_1.weight"))


def encode_text(tokens: torch.Tensor, sd: SD, heads: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """CLIP.encode_text, reference model/model.py:190-201: embedding + positional,
    causal blocks, ln_final, row at argmax(token id) (= EOT), @ text_projection."""
    sd = _cast(sd, dtype)
    x = sd["token_embedding.weight"][tokens.long()] + sd["positional_embedding"]
    mask = causal_mask(x.shape[1], dtype)
    for i in range(_text_layers(sd)):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", heads, mask)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = tokens.long().argmax(dim=-1)
    return x[torch.arange(x.shape[0]), eot] @ sd["text_projection"]


def adapted_encode_text(tokens: torch.Tensor, sd: SD, ta: SD, heads: int,
                        text_adapt_weight: float = 0.1, text_adapt_until: int = 3,
                        dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """AdaptedCLIP.encode_text(text, adapt_text=True), reference
    model/adapter.py:273-304: adapters after blocks < until, ln_final, EOT row,
    then text_adapter[-1] (Linear no-bias + LeakyReLU) replaces text_projection."""
    sd, ta = _cast(sd, dtype), _cast(ta, dtype)
    x = sd["token_embedding.weight"][tokens.long()] + sd["positional_embedding"]
    mask = causal_mask(x.shape[1], dtype)
    for i in range(_text_layers(sd)):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", heads, mask)
        if i < text_adapt_until:
            x = adapter_mix(x, ta[f"{i}.fc.0.weight"], text_adapt_weight)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = tokens.long().argmax(dim=-1)
    x = x[torch.arange(x.shape[0]), eot]
    return leaky_relu(x @ ta[f"{text_adapt_until}.fc.0.weight"].t())


def class_anchor(normal_emb: torch.Tensor, abnormal_emb: torch.Tensor) -> torch.Tensor:
    """reference forward_utils.py:154-161: per state, L2-normalise sentence rows,
    mean over sentences, L2-normalise, stack -> [E, 2] (col 0 normal, col 1 abnormal)."""
    cols = []
    for e in (normal_emb, abnormal_emb):
        e = e / e.norm(dim=-1, keepdim=True)
        m = e.mean(dim=0)
        cols.append(m / m.norm())
    return torch.stack(cols, dim=1)


# --------------------------------------------------------------------------
# anomaly map
# --------------------------------------------------------------------------
def gaussian_kernel1d(ksize: int, sigma: float, dtype: torch.dtype) -> torch.Tensor:
    """kornia 0.6.9 filters.get_gaussian_kernel1d: x = arange(k) - k//2
    (+0.5 if k even), exp(-x^2 / (2 sigma^2)), normalised to sum 1."""
    x = torch.arange(ksize, dtype=dtype) - ksize // 2
    if ksize % 2 == 0:
        x = x + 0.5
    g = torch.exp(-(x ** 2) / (2.0 * sigma * sigma))
    return g / g.sum()


def gaussian_blur2d(x: torch.Tensor, ksize: int, sigma: float) -> torch.Tensor:
    """kornia 0.6.9 gaussian_blur2d(input, (k,k), (s,s)) defaults: border_type
    'reflect', separable.  PARITY UNPINNED (kornia not installed here); restated
    from its published algorithm.  x is [B, C, H, W]."""
    B, C, H, W = x.shape
    k = gaussian_kernel1d(ksize, sigma, x.dtype)
    r = ksize // 2
    xp = F.pad(x, (r, r, r, r), mode="reflect")
    xp = xp.reshape(B * C, 1, H + 2 * r, W + 2 * r)
    xp = F.conv2d(xp, k.view(1, 1, 1, ksize))
    xp = F.conv2d(xp, k.view(1, 1, ksize, 1))
    return xp.reshape(B, C, H, W)


def bilinear_align_corners(x: torch.Tensor, size: int) -> torch.Tensor:
    """F.interpolate(mode='bilinear', align_corners=True), written out:
    src = dst * (in-1)/(out-1); used at reference forward_utils.py:211-213."""
    B, C, H, W = x.shape

    def axis(n_in: int):
        scale = (n_in - 1) / (size - 1) if size > 1 else 0.0
        pos = torch.arange(size, dtype=x.dtype) * scale
        i0 = pos.floor().clamp(max=n_in - 1).long()
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, pos - i0.to(x.dtype)

    y0, y1, fy = axis(H)
    x0, x1, fx = axis(W)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy).view(1, 1, -1, 1) + bot * fy.view(1, 1, -1, 1)


def similarity_map(patch_features: torch.Tensor, text_feature: torch.Tensor, img_size: int,
                   test: bool = False, domain: str = "Medical") -> torch.Tensor:
    """calculate_similarity_map, reference forward_utils.py:196-216.
    patch_features [B,P,E]; text_feature [E,2] or [B,E,2].
    test=True -> [B,1,S,S]; test=False -> softmax over the 2 channels, [B,2,S,S]."""
    s = 100.0 * torch.matmul(patch_features, text_feature)
    B, L, C = s.shape
    g = int(math.isqrt(L))
    m = s.permute(0, 2, 1).reshape(B, C, g, g)
    if test:
        assert C == 2
        sigma, k = (1.0, 7) if domain == "Industrial" else (1.5, 9)
        m = ((m[:, 1] + 1 - m[:, 0]) / 2).unsqueeze(1)
        m = gaussian_blur2d(m, k, sigma)
    m = bilinear_align_corners(m, img_size)
    if not test and C > 1:
        m = torch.softmax(m, dim=1)
    return m


def anomaly_map(seg_tokens: Sequence[torch.Tensor], text_feature: torch.Tensor, img_size: int,
                domain: str = "Industrial") -> torch.Tensor:
    """Text-only branch of get_predictions, reference test_last.py:95-100,149:
    sum over the tap levels of the test-mode similarity maps -> [B,S,S]."""
    maps = [similarity_map(f, text_feature, img_size, test=True, domain=domain) for f in seg_tokens]
    return torch.cat(maps, dim=1).sum(dim=1)


def image_score(det_token: torch.Tensor, text_feature: torch.Tensor) -> torch.Tensor:
    """Intended image-level score (det_b . t_abnormal + 1)/2 -> [B].  The
    reference line test_last.py:90-91 broadcasts [B,E]@[B,E,2] to [B,B,2] and
    picks row 1; the build computes the intended per-image
    score and documents the deviation."""
    t = text_feature if text_feature.dim() == 2 else text_feature[0]
    return (det_token @ t[:, 1] + 1) / 2


def image_score_reference_quirk(det_token: torch.Tensor, text_feature_b: torch.Tensor) -> torch.Tensor:
    """Bit-for-bit what reference test_last.py:90-91 evaluates (for the record):
    pred = det[B,E] @ t[B,E,2] -> [B,B,2]; (pred[:,1]+1)/2 -> [B,2]."""
    pred = det_token @ text_feature_b
    return (pred[:, 1] + 1) / 2


# --------------------------------------------------------------------------
# load-time positional-embedding resize
# --------------------------------------------------------------------------
def resize_pos_embed(pos: torch.Tensor, new_grid: int) -> torch.Tensor:
    """reference model/model.py:396-427: keep the CLS row, bicubic + antialias
    (align_corners=False) resize of the [g,g,D] grid to [new_grid,new_grid,D]."""
    tok, img = pos[:1], pos[1:]
    g = int(math.isqrt(img.shape[0]))
    if g == new_grid:
        return pos
    img = img.reshape(1, g, g, -1).permute(0, 3, 1, 2)
    img = F.interpolate(img, size=(new_grid, new_grid), mode="bicubic", antialias=True, align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(new_grid * new_grid, -1)
    return torch.cat([tok, img], dim=0)
