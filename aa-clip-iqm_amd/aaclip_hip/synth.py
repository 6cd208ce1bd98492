"""Deterministic synthetic weights for the AA-CLIP hot path.

There are no pretrained checkpoints in the container or on the GPU box, so every
test, the smoke run and bench.py use weights regenerated from a seed.  Each
tensor gets its own torch CPU generator seeded from crc32(name) ^ seed, so the
same name always yields the same values on every machine running this image,
independent of the order tensors are created in.

The standard deviations follow the reference initialisers so activations have
realistic magnitudes (reference model/transformer.py:606-623: attn D^-1/2,
proj (2*layers)^-1/2 * D^-1/2, fc (2D)^-1/2; adapters xavier-uniform,
reference model/adapter.py:107-113).  LayerNorm gains/biases and linear biases
are made non-trivial on purpose so those code paths are exercised, and the
q/k rows of in_proj are scaled x2 so softmax rows are not near-uniform.

The key names and shapes are the reference's state-dict contract
(reference model/model.py:311-369, strict load at model/clip.py:132).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List

import torch


@dataclass
class TowerCfg:
    width: int
    layers: int
    heads: int
    mlp: int


@dataclass
class ClipCfg:
    """Architecture numbers of reference model/model_configs/ViT-L-14-336.json
    (image_size overridden to 518 by create_model, model/clip.py:112)."""
    embed_dim: int = 768
    image_size: int = 518
    patch_size: int = 14
    vision: TowerCfg = field(default_factory=lambda: TowerCfg(1024, 24, 16, 4096))
    text: TowerCfg = field(default_factory=lambda: TowerCfg(768, 12, 12, 3072))
    context_length: int = 77
    vocab_size: int = 49408

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1


def tiny_cfg() -> ClipCfg:
    """Reduced architecture for fast tests: same structure, head dim 64."""
    return ClipCfg(
        embed_dim=256,
        image_size=70,
        patch_size=14,
        vision=TowerCfg(256, 3, 4, 1024),
        text=TowerCfg(256, 2, 4, 1024),
        context_length=77,
        vocab_size=49408,
    )


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def randn(name: str, shape, std: float, seed: int, mean: float = 0.0) -> torch.Tensor:
    t = torch.randn(*shape, generator=_gen(name, seed), dtype=torch.float32)
    return t.mul_(std).add_(mean)


def uniform(name: str, shape, bound: float, seed: int) -> torch.Tensor:
    t = torch.rand(*shape, generator=_gen(name, seed), dtype=torch.float32)
    return t.mul_(2 * bound).sub_(bound)


def _tower(sd: Dict[str, torch.Tensor], prefix: str, cfg: TowerCfg, seed: int) -> None:
    d = cfg.width
    attn_std = d ** -0.5
    proj_std = (d ** -0.5) * ((2 * cfg.layers) ** -0.5)
    fc_std = (2 * d) ** -0.5
    for i in range(cfg.layers):
        p = f"{prefix}resblocks.{i}."
        sd[p + "ln_1.weight"] = randn(p + "ln_1.weight", (d,), 0.1, seed, 1.0)
        sd[p + "ln_1.bias"] = randn(p + "ln_1.bias", (d,), 0.05, seed)
        w = randn(p + "attn.in_proj_weight", (3 * d, d), attn_std, seed)
        w[: 2 * d].mul_(2.0)
        sd[p + "attn.in_proj_weight"] = w
        sd[p + "attn.in_proj_bias"] = randn(p + "attn.in_proj_bias", (3 * d,), 0.02, seed)
        sd[p + "attn.out_proj.weight"] = randn(p + "attn.out_proj.weight", (d, d), proj_std, seed)
        sd[p + "attn.out_proj.bias"] = randn(p + "attn.out_proj.bias", (d,), 0.02, seed)
        sd[p + "ln_2.weight"] = randn(p + "ln_2.weight", (d,), 0.1, seed, 1.0)
        sd[p + "ln_2.bias"] = randn(p + "ln_2.bias", (d,), 0.05, seed)
        sd[p + "mlp.c_fc.weight"] = randn(p + "mlp.c_fc.weight", (cfg.mlp, d), fc_std, seed)
        sd[p + "mlp.c_fc.bias"] = randn(p + "mlp.c_fc.bias", (cfg.mlp,), 0.02, seed)
        sd[p + "mlp.c_proj.weight"] = randn(p + "mlp.c_proj.weight", (d, cfg.mlp), proj_std, seed)
        sd[p + "mlp.c_proj.bias"] = randn(p + "mlp.c_proj.bias", (d,), 0.02, seed)


def synth_clip_state_dict(cfg: ClipCfg, seed: int = 111) -> Dict[str, torch.Tensor]:
    """Full CLIP state dict with the reference's key names (positional
    embedding already at the run-time grid, i.e. after resize_pos_embed)."""
    sd: Dict[str, torch.Tensor] = {}
    dv, dt = cfg.vision.width, cfg.text.width
    ps = cfg.patch_size
    sd["visual.conv1.weight"] = randn("visual.conv1.weight", (dv, 3, ps, ps), (3 * ps * ps) ** -0.5, seed)
    sd["visual.class_embedding"] = randn("visual.class_embedding", (dv,), dv ** -0.5, seed)
    sd["visual.positional_embedding"] = randn("visual.positional_embedding", (cfg.tokens, dv), dv ** -0.5, seed)
    sd["visual.ln_pre.weight"] = randn("visual.ln_pre.weight", (dv,), 0.1, seed, 1.0)
    sd["visual.ln_pre.bias"] = randn("visual.ln_pre.bias", (dv,), 0.05, seed)
    _tower(sd, "visual.transformer.", cfg.vision, seed)
    sd["visual.ln_post.weight"] = randn("visual.ln_post.weight", (dv,), 0.1, seed, 1.0)
    sd["visual.ln_post.bias"] = randn("visual.ln_post.bias", (dv,), 0.05, seed)
    sd["visual.proj"] = randn("visual.proj", (dv, cfg.embed_dim), dv ** -0.5, seed)
    sd["token_embedding.weight"] = randn("token_embedding.weight", (cfg.vocab_size, dt), 0.02, seed)
    sd["positional_embedding"] = randn("positional_embedding", (cfg.context_length, dt), 0.01, seed)
    _tower(sd, "transformer.", cfg.text, seed)
    sd["ln_final.weight"] = randn("ln_final.weight", (dt,), 0.1, seed, 1.0)
    sd["ln_final.bias"] = randn("ln_final.bias", (dt,), 0.05, seed)
    sd["text_projection"] = randn("text_projection", (dt, cfg.embed_dim), dt ** -0.5, seed)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    return sd


# Channels of the visual residual stream that `outlier_edit` turns into "massive activation" channels, with the constant
# each receives from block 6's MLP bias (real ViT-L/14 checkpoints carry a handful of such channels in the hundreds).
OUTLIER_CHANNELS = ((7, 120.0), (133, -80.0), (402, 300.0), (518, -500.0), (777, 600.0), (1001, -150.0))
OUTLIER_FROM_BLOCK = 5            # 0-based: the stream carries the outliers from the output of the 6th block on
OUTLIER_GELU_UNITS = {8: ((100, 700.0), (2000, 1500.0)), 15: ((3500, 2500.0), (64, 460.0))}   # block -> (hidden unit, c_fc bias)


def outlier_edit(sd: Dict[str, torch.Tensor], cfg: "ClipCfg", seed: int = 111) -> Dict[str, torch.Tensor]:
    """A copy of a synthetic CLIP state dict edited so that the VISUAL tower behaves like a trained checkpoint with
    outlier channels (all other parity records are Gaussian-initialised weights):

      * block 6's `mlp.c_proj` writes constants of +-80 ... +-600 into six channels of the residual stream (bias) and
        its rows for those channels are 6x larger, so the outliers also vary by token; block 12's `attn.out_proj` does
        the same for two more channels at +-40.  From there on every LayerNorm statistic is dominated by them.
      * every LayerNorm behind them (ln_1 / ln_2 of blocks 7..24, ln_post) has gamma x5..x10 on the ordinary channels
        (a trained model compensates the shrunken normalised values this way) and gamma x0.02..x0.1 on the outlier channels.
      * blocks 9 and 16 have hidden units whose pre-activation sits at 460 ... 2500, i.e. GELU outputs beyond the
        e4m3 range (448) of the split8 correction planes, feeding `c_proj` columns scaled by 0.02.

    Deterministic (per-tensor generators like every other synthetic weight); used by tests/golden/make_golden_full4o.py
    for the reference run and by tests/test_gpu_configs.py for the build's run of the same weights."""
    out = {k: v.clone() for k, v in sd.items()}
    d, layers = cfg.vision.width, cfg.vision.layers
    pre = "visual.transformer.resblocks."
    ch = torch.tensor([c for c, _ in OUTLIER_CHANNELS])
    val = torch.tensor([v for _, v in OUTLIER_CHANNELS])
    b = OUTLIER_FROM_BLOCK
    out[f"{pre}{b}.mlp.c_proj.bias"][ch] += val
    out[f"{pre}{b}.mlp.c_proj.weight"][ch] *= 6.0
    ch2 = torch.tensor([55, 640])
    out[f"{pre}11.attn.out_proj.bias"][ch2] += torch.tensor([40.0, -40.0])
    out[f"{pre}11.attn.out_proj.weight"][ch2] *= 8.0
    allch = torch.cat([ch, ch2])

    def regain(name):
        g = 5.0 + 5.0 * torch.rand(d, generator=_gen("outlier." + name, seed))
        g[allch] = 0.02 + 0.08 * torch.rand(allch.numel(), generator=_gen("outlier.small." + name, seed))
        out[name] *= g

    for i in range(b + 1, layers):
        regain(f"{pre}{i}.ln_1.weight")
        regain(f"{pre}{i}.ln_2.weight")
    regain("visual.ln_post.weight")
    for blk, units in OUTLIER_GELU_UNITS.items():
        for j, bias in units:
            out[f"{pre}{blk}.mlp.c_fc.bias"][j] = bias
            out[f"{pre}{blk}.mlp.c_fc.weight"][j] *= 4.0
            out[f"{pre}{blk}.mlp.c_proj.weight"][:, j] *= 0.02
    return out


def _xavier(name: str, out_f: int, in_f: int, seed: int) -> torch.Tensor:
    return uniform(name, (out_f, in_f), math.sqrt(6.0 / (in_f + out_f)), seed)


def synth_image_adapter_state_dict(cfg: ClipCfg, until: int = 6, levels: int = 4,
                                   relu: bool = False, seed: int = 111) -> Dict[str, torch.Tensor]:
    """Keys of AdaptedCLIP.image_adapter (reference model/adapter.py:35-48)."""
    dv, e = cfg.vision.width, cfg.embed_dim
    proj_key = "fc.0.weight" if relu else "fc.weight"
    sd = {}
    for i in range(until):
        k = f"layer_adapters.{i}.fc.0.weight"
        sd[k] = _xavier("image_adapter." + k, dv, dv, seed)
    for i in range(levels):
        k = f"seg_proj.{i}.{proj_key}"
        sd[k] = _xavier("image_adapter." + k, e, dv, seed)
    k = f"det_proj.{proj_key}"
    sd[k] = _xavier("image_adapter." + k, e, dv, seed)
    return sd


def synth_text_adapter_state_dict(cfg: ClipCfg, until: int = 3, seed: int = 111) -> Dict[str, torch.Tensor]:
    """Keys of AdaptedCLIP.text_adapter (reference model/adapter.py:51-54):
    `until` SimpleAdapter(width,width) then one SimpleProj(width, embed, relu=True)."""
    dt, e = cfg.text.width, cfg.embed_dim
    sd = {}
    for i in range(until):
        k = f"{i}.fc.0.weight"
        sd[k] = _xavier("text_adapter." + k, dt, dt, seed)
    k = f"{until}.fc.0.weight"
    sd[k] = _xavier("text_adapter." + k, e, dt, seed)
    return sd


def synth_iqm_state_dict(cfg: ClipCfg, levels: int = 4, relu: bool = False, hidden: int = 768, layers: int = 2,
                         inter: int = 2048, seed: int = 111) -> Dict[str, torch.Tensor]:
    """Seeded weights of everything the IQM side branch owns, under the reference's parameter names relative to
    AdaptedCLIP (reference model/adapter.py:56-92, model/iqm.py): `iqm.*`, `class_query_mlp.{0,2}.*`,
    `query_adapters.{i}.fc[.0].weight`, `iqm_layer_norm.*`, `pos_embedding`, `visual_weight`, `text_weight`, plus the
    two projections the reference creates lazily with fresh random weights on the first forward and never saves
    (`visual_feature_proj` Linear(hidden, hidden), adapter.py:213-218; `text_feature_proj` Linear(2, 768) -- 2 because
    the anchors arrive as [B, 768, 2], adapter.py:229-243): here they are ordinary seeded parameters.
    Linear weights xavier-uniform like the reference's _init_weights_ (:107-123), biases and LayerNorm affine
    parameters non-trivial so those paths are exercised."""
    dv = cfg.vision.width
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=True):
        sd[name + ".weight"] = _xavier("iqm_branch." + name, out_f, in_f, seed)
        if bias:
            sd[name + ".bias"] = randn("iqm_branch." + name + ".bias", (out_f,), 0.05, seed)

    def ln(name, d):
        sd[name + ".weight"] = randn("iqm_branch." + name + ".w", (d,), 0.1, seed, 1.0)
        sd[name + ".bias"] = randn("iqm_branch." + name + ".b", (d,), 0.05, seed)

    lin("class_query_mlp.0", hidden, dv)
    lin("class_query_mlp.2", hidden, hidden)
    for i in range(levels):
        k = f"query_adapters.{i}." + ("fc.0" if relu else "fc")
        sd[k + ".weight"] = _xavier("iqm_branch." + k, hidden, dv, seed)
    lin("visual_feature_proj", hidden, hidden)
    lin("text_feature_proj", 768, 2)
    ln("iqm_layer_norm", hidden)
    ln("iqm.layernorm", hidden)
    for l in range(layers):
        p = f"iqm.encoder.layer.{l}."
        for att in ("attention", "crossattention", "text_crossattention"):
            for m in ("query", "key", "value"):
                lin(p + att + ".attention." + m, hidden, hidden)
            lin(p + att + ".output.dense", hidden, hidden)
            ln(p + att + ".output.LayerNorm", hidden)
        for suffix in ("", "_query"):
            lin(p + "intermediate" + suffix + ".dense", inter, hidden)
            lin(p + "output" + suffix + ".dense", hidden, inter)
            ln(p + "output" + suffix + ".LayerNorm", hidden)
    pos = torch.arange(512, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, hidden, 2, dtype=torch.float32) * (-math.log(10000.0) / hidden))
    pe = torch.zeros(512, hidden)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    sd["pos_embedding"] = pe.unsqueeze(0)          # [1, 512, hidden], reference adapter.py:98-105
    sd["visual_weight"] = torch.tensor(0.6)
    sd["text_weight"] = torch.tensor(0.4)
    return sd


def synth_images(batch: int, size: int, seed: int = 111, offset: int = 0) -> torch.Tensor:
    """CLIP-normalised pixels are ~N(0,1); image i only depends on
    (seed, offset+i) so any rank can regenerate its shard."""
    out = torch.empty(batch, 3, size, size, dtype=torch.float32)
    for i in range(batch):
        out[i] = torch.randn(3, size, size, generator=_gen(f"image.{offset + i}", seed))
    return out


def synth_masks(batch: int, size: int, seed: int = 111, offset: int = 0) -> torch.Tensor:
    """Synthetic binary ground-truth masks (random rectangles/ellipses, ~5 %
    positive) for AUROC parity when no dataset is present."""
    out = torch.zeros(batch, size, size, dtype=torch.uint8)
    ys = torch.arange(size).view(-1, 1).float()
    xs = torch.arange(size).view(1, -1).float()
    for i in range(batch):
        g = _gen(f"mask.{offset + i}", seed)
        r = torch.rand(6, generator=g)
        cy, cx = (0.15 + 0.7 * r[0]) * size, (0.15 + 0.7 * r[1]) * size
        hh, hw = (0.05 + 0.12 * r[2]) * size, (0.05 + 0.12 * r[3]) * size
        if r[4] < 0.5:
            m = ((ys - cy).abs() <= hh) & ((xs - cx).abs() <= hw)
        else:
            m = ((ys - cy) / hh) ** 2 + ((xs - cx) / hw) ** 2 <= 1.0
        out[i] = m.to(torch.uint8)
    return out
