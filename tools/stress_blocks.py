#!/usr/bin/env python3
"""Randomised sweep of aaclip_blocks (widths, lengths, batch sizes on both sides of the large-batch kernels, causal /
full / V-V attention, adapters, runs of 1-3 blocks, fp16 / bf16 / fp16x2, taps through aaclip_blocks_to) against an fp64 torch
restatement on the GPU.  One-off confidence run.  usage: python tools/stress_blocks.py [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import numpy as np
import torch
from aaclip_hip import engine
from aaclip_hip._lib import F16, BF16, F16X2
from model.transformer import ResidualAttentionBlock

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
codes = {"all": [F16, BF16, F16X2], "fp16x2": [F16X2]}[sys.argv[2] if len(sys.argv) > 2 else "all"]   # usage: seed [all|fp16x2]
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
fails = 0


def ln(x, w, b):
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    return (x - m) / torch.sqrt(v + 1e-5) * w + b


def ref_block(x, blk, B, L, H, causal, vv, aw, mix):
    D = x.shape[1]
    d = lambda t: t.detach().double()
    h = ln(x, d(blk.ln_1.weight), d(blk.ln_1.bias))
    qkv = (h @ d(blk.attn.in_proj_weight).t() + d(blk.attn.in_proj_bias)).view(B, L, 3, H, 64)
    if vv:   # attention over the batch axis with q = k = v
        v = qkv[:, :, 2].permute(1, 2, 0, 3)                       # [L, H, B, 64]
        p = torch.softmax(v @ v.transpose(-1, -2) * 0.125, -1)
        ctx = (p @ v).permute(2, 0, 1, 3).reshape(B * L, D)
    else:
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        s = q @ k.transpose(-1, -2) * 0.125
        if causal:
            s = s + torch.full((L, L), float("-inf"), device=x.device, dtype=torch.float64).triu_(1)
        ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, D)
    x = x + ctx @ d(blk.attn.out_proj.weight).t() + d(blk.attn.out_proj.bias)
    h = ln(x, d(blk.ln_2.weight), d(blk.ln_2.bias))
    h = h @ d(blk.mlp.c_fc.weight).t() + d(blk.mlp.c_fc.bias)
    h = 0.5 * h * (1 + torch.erf(h / 2 ** 0.5))
    x = x + h @ d(blk.mlp.c_proj.weight).t() + d(blk.mlp.c_proj.bias)
    if aw is not None:
        a = torch.nn.functional.leaky_relu(x @ d(aw).t(), 0.01)
        a = a * x.norm(dim=-1, keepdim=True) / a.norm(dim=-1, keepdim=True)
        x = mix * a + (1 - mix) * x
    return x


n = 0
for _ in range(10):
    code = codes[int(rng.integers(len(codes)))]
    D = int(rng.choice([256, 512, 768, 1024])); H = D // 64
    L = int(rng.choice([26, 77, 200, 513, 1370]))
    big = bool(rng.integers(2))
    B = max(1, (4500 // L + 1) if big else max(1, 2000 // L))
    B = min(B, 40)
    nblk = int(rng.integers(1, 4))
    mode = int(rng.choice([0, 0, 1, 2]))   # full, causal, V-V
    causal, vv = mode == 1, mode == 2
    use_ad = bool(rng.integers(2)) and not vv
    blocks = []
    for _b in range(nblk):
        blk = ResidualAttentionBlock(D, H).to(dev)
        with torch.no_grad():
            for p in blk.parameters():
                if p.dim() > 1:
                    p.normal_(0, 0.7 * p.shape[1] ** -0.5)
                else:
                    p.normal_(0, 0.3)
            blk.ln_1.weight.add_(1.0); blk.ln_2.weight.add_(1.0)
        if vv:
            blk.surgery = True
        blocks.append(blk)
    aws = [(torch.randn(D, D, device=dev) * D ** -0.5) if use_ad else None for _ in blocks]
    x0 = torch.randn(B * L, D, device=dev)
    x0[:, 3] += 2.0
    xa = x0.clone()
    tap = bool(rng.integers(2))
    with torch.no_grad():
        if tap:   # read x0, continue in a fresh buffer
            out = torch.empty_like(xa)
            engine.run_blocks(xa, blocks, B, L, H, code, causal=causal, adapter_weights=aws, mix=0.1, x_out=out)
            if not torch.equal(xa, x0):
                fails += 1
                print("FAIL tapped buffer was modified")
            xa = out
        else:
            engine.run_blocks(xa, blocks, B, L, H, code, causal=causal, adapter_weights=aws, mix=0.1)
        ref = x0.double()
        for blk, aw in zip(blocks, aws):
            ref = ref_block(ref, blk, B, L, H, causal, vv, aw, 0.1)
    err = (xa.double() - ref).abs()
    # fp16x2: up to 3 blocks of unit-scale random weights; plain fp16 is asserted 10x looser
    atol, rtol = {F16: (1.2e-2, 1.5e-2), F16X2: (1.2e-3, 1.5e-3)}.get(code, (8e-2, 6e-2))
    bad = err > atol + rtol * ref.abs()
    tag = f"D{D} L{L} B{B} M{B*L} blocks{nblk} mode{mode} adapter{int(use_ad)} tap{int(tap)} code{code}"
    if bad.any() or not torch.isfinite(xa).all():
        fails += 1
        print(f"FAIL {tag}: {int(bad.sum())}/{bad.numel()} outside, max err {err.max().item():.3e} (ref max {ref.abs().max().item():.1f})")
    else:
        print(f"ok   {tag}: max err {err.max().item():.2e}")
    n += 1
    del blocks, aws, x0, xa, ref
print("FAILURES:", fails)
sys.exit(1 if fails else 0)
