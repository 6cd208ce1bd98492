"""Parameter containers with the reference's module tree (so state dicts are
interchangeable, reference model/transformer.py) whose forward methods call the
HIP path in aaclip_hip.engine.  Nothing here computes with torch ops.

Layout note: the reference runs its blocks on LND tensors; the engine keeps the
residual stream batch-first ([B*L, D] fp32).  The block / transformer modules
accept LND input for API compatibility and convert at the boundary; the fused
entry points (CLIP.encode_image / encode_text, AdaptedCLIP.forward /
encode_text) never leave the batch-first layout.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch import nn

from aaclip_hip import engine


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class LayerNorm(nn.Module):
    """nn.LayerNorm replacement (reference model/transformer.py:37-43); forward
    runs the wavefront-reduction HIP kernel and returns fp32 like the input."""

    def __init__(self, width: int, eps: float = 1e-5):
        super().__init__()
        self.normalized_shape = (width,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return engine.layernorm(x, self.weight, self.bias, self.eps).to(x.dtype)


LayerNormFp32 = LayerNorm


class Linear(nn.Module):
    """Holder for an nn.Linear weight/bias pair; a direct call goes through the
    HIP GEMM (fp32 result)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, std: Optional[float] = None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.normal_(self.weight, std=std if std is not None else in_features ** -0.5)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return engine.linear(x, self.weight, self.bias, False, engine.dtype_code(getattr(self, "precision", "fp32")))


class MultiheadAttentionParams(nn.Module):
    """Packed-projection parameters with nn.MultiheadAttention's key names
    (reference model/transformer.py:200): in_proj_weight/bias, out_proj.*"""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)
        nn.init.normal_(self.in_proj_weight, std=d_model ** -0.5)


class Mlp(nn.Module):
    def __init__(self, d_model: int, mlp_width: int):
        super().__init__()
        self.c_fc = Linear(d_model, mlp_width, std=(2 * d_model) ** -0.5)
        self.gelu = nn.Identity()  # exact-erf GELU is fused into the c_fc GEMM epilogue
        self.c_proj = Linear(mlp_width, d_model)


class ResidualAttentionBlock(nn.Module):
    """reference model/transformer.py:183-258 (pre-LN, ls_1/ls_2 identity)."""

    def __init__(self, d_model: int, n_head: int, mlp_ratio: float = 4.0, idx: int = 12):
        super().__init__()
        if d_model != 64 * n_head:
            raise ValueError("the HIP attention kernel is built for head_dim 64 (d_model == 64 * n_head)")
        self.idx = idx
        self.n_head = n_head
        self.ln_1 = LayerNorm(d_model)
        self.attn = MultiheadAttentionParams(d_model, n_head)
        self.ln_2 = LayerNorm(d_model)
        self.mlp = Mlp(d_model, int(d_model * mlp_ratio))

    def forward(self, q_x: torch.Tensor, k_x=None, v_x=None, attn_mask: Optional[torch.Tensor] = None):
        """LND in / LND out, returns (x, None): the head-averaged attention
        weights the reference also returns (transformer.py:258) are never
        materialised by the fused kernel and no caller on the path reads them.
        A non-None attn_mask must be the causal mask of transformer.py:629-635."""
        if k_x is not None or v_x is not None:
            raise NotImplementedError("cross-attention inputs are outside the AA-CLIP hot path")
        L, B, D = q_x.shape
        x = q_x.detach().permute(1, 0, 2).float().contiguous().view(B * L, D)
        code = engine.dtype_code(getattr(self, "precision", "fp32"))
        engine.run_block(x, self, B, L, self.n_head, code, causal=attn_mask is not None)
        return x.view(B, L, D).permute(1, 0, 2).to(q_x.dtype), None


class Transformer(nn.Module):
    """reference model/transformer.py:261-317."""

    def __init__(self, width: int, layers: int, heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.grad_checkpointing = False
        self.resblocks = nn.ModuleList(
            [ResidualAttentionBlock(width, heads, mlp_ratio, idx=i) for i in range(layers)])
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        for blk in self.resblocks:
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)

    def get_cast_dtype(self) -> torch.dtype:
        return self.resblocks[0].mlp.c_fc.weight.dtype

    def run(self, x: torch.Tensor, B: int, L: int, code: int, causal: bool, out_layers: Sequence[int] = (),
            adapter_weights: Optional[Sequence[Optional[torch.Tensor]]] = None, mix: float = 0.0):
        """Batch-first tower -> (final stream, [stream after the 1-based layers in out_layers]) (reference
        transformer.py:295-317).  Without taps the tower runs in place on x; a tapped buffer is never written
        again (the next run continues in a fresh one), so taps cost no copy and the final stream may live in a
        different tensor than x.  The library sees the whole tower (aaclip_blocks_taps), so ln_1 is folded into the
        QKV product behind a tap as well.  adapter_weights[i] (or None) = the residual adapter applied after block i
        (reference model/adapter.py:163-170), mixed in with weight `mix`."""
        taps = []
        run: list = []     # consecutive blocks of one attention mode: ONE aaclip_blocks_taps call, taps included
        outs: list = []    # per block of the run: the buffer holding the stream after it
        n = len(self.resblocks)
        cur = x            # buffer holding the stream; after a tap the next block continues in a fresh one
        src = x            # what the run's first block reads
        first = 0          # index of the run's first block
        tapped = False
        for i, blk in enumerate(self.resblocks):
            if tapped:     # `cur` was handed out as a tap: read it, write the continuation elsewhere (no copy)
                cur, tapped = torch.empty_like(cur), False
            run.append(blk)
            outs.append(cur)
            nxt = self.resblocks[i + 1] if i + 1 < n else None
            if nxt is None or bool(getattr(nxt, "surgery", False)) != bool(getattr(blk, "surgery", False)):
                aws = adapter_weights[first:i + 1] if adapter_weights is not None else None
                engine.run_blocks(src, run, B, L, self.heads, code, causal=causal, adapter_weights=aws, mix=mix,
                                  x_outs=outs)
                run, outs, src, first = [], [], cur, i + 1
            if (i + 1) in out_layers:
                taps.append(cur)
                tapped = True
        return cur, taps

    def forward(self, x: torch.Tensor, out_layers: list = [3, 6, 9], attn_mask: Optional[torch.Tensor] = None):
        L, B, D = x.shape
        xb = x.detach().permute(1, 0, 2).float().contiguous().view(B * L, D)
        code = engine.dtype_code(getattr(self, "precision", "fp32"))
        xb, taps = self.run(xb, B, L, code, attn_mask is not None, out_layers)
        lnd = lambda t: t.view(B, L, D).permute(1, 0, 2)
        return lnd(xb), [lnd(t) for t in taps]


class PatchConv(nn.Module):
    """Holder of conv1.weight [D,3,ps,ps] (reference transformer.py:359-365); the
    convolution itself is the im2col-free GEMM inside aaclip_patch_embed."""

    def __init__(self, width: int, patch_size: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch_size, patch_size))
        nn.init.normal_(self.weight, std=(3 * patch_size * patch_size) ** -0.5)

    def forward(self, x):
        raise NotImplementedError(
            "conv1 is fused with class/positional embedding and ln_pre in aaclip_patch_embed; "
            "call CLIP.encode_image or AdaptedCLIP.forward")


class VisionTransformer(nn.Module):
    """reference model/transformer.py:320-551 (ViT branch, no attentional pool)."""

    def __init__(self, image_size: int, patch_size: int, width: int, layers: int, heads: int, mlp_ratio: float,
                 output_dim: int = 512, patch_dropout: float = 0.0, **_unused):
        super().__init__()
        self.image_size = to_2tuple(image_size)
        self.patch_size = to_2tuple(patch_size)
        self.grid_size = (self.image_size[0] // self.patch_size[0], self.image_size[1] // self.patch_size[1])
        self.output_dim = output_dim
        self.embed_dim, self.num_heads = width, heads
        self.conv1 = PatchConv(width, self.patch_size[0])
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.patch_dropout = nn.Identity()  # eval-mode identity (reference transformer.py:74)
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.global_average_pool = False
        self.attn_pool = None

    @torch.no_grad()
    def DAPM_replace(self, DPAM_layer):
        """reference transformer.py:406-425: the last DPAM_layer-1 blocks get the V-V "surgery"
        attention (:102-152) with the block's own in_proj/out_proj weights.  No module is swapped
        here: the block is flagged and aaclip_block runs AACLIP_ATTN_VV_BATCH, which reproduces what
        the reference executes -- attention over the batch axis per token position (SURVEY.md 8(f)
        F3), so features depend on which images share a batch, exactly as in the reference."""
        if DPAM_layer is not None:
            for i in range(1, DPAM_layer):
                self.transformer.resblocks[-i].surgery = True

    def _global_pool(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return x[:, 0], x[:, 1:]

    def forward(self, x: torch.Tensor, out_layers: list):
        """reference transformer.py:490-551 -> (pooled [B,E], [tokens [B,L,D] at out_layers])."""
        code = engine.dtype_code(getattr(self, "precision", "fp32"))
        xs, B, L = engine.patch_embed(x, self, code)
        xs, taps = self.transformer.run(xs, B, L, code, False, out_layers)
        pooled = engine.row_head(xs, None, self.ln_post, self.proj, "transpose", False, B, L, 1, code)
        D = self.embed_dim
        return pooled, [t.view(B, L, D) for t in taps]


def causal_attn_mask(n: int) -> torch.Tensor:
    """reference transformer.py:629-635."""
    return torch.full((n, n), float("-inf")).triu_(1)


def set_precision(module: nn.Module, precision) -> None:
    for m in module.modules():
        m.precision = precision
