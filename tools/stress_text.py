#!/usr/bin/env python3
"""Randomised sweep of encode_text (plain and adapted) on the reduced model: random token ids, end-of-text position
anywhere in the 77 slots, 1..300 sentences (both GEMM kernel families), fp16, against the fp64 oracle.
usage: python tools/stress_text.py [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from aaclip_hip import synth
from oracle import aaclip_oracle as O
from model.model import CLIP
from model.adapter import AdaptedCLIP

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
cfg = synth.tiny_cfg()
sd = synth.synth_clip_state_dict(cfg, seed=7)
clip = CLIP(cfg.embed_dim, dict(image_size=cfg.image_size, layers=cfg.vision.layers, width=cfg.vision.width,
                                patch_size=cfg.patch_size),
            dict(context_length=77, vocab_size=cfg.vocab_size, width=cfg.text.width, heads=cfg.text.heads,
                 layers=cfg.text.layers), precision="fp16")
clip.load_state_dict(sd, strict=True)
ta = synth.synth_text_adapter_state_dict(cfg, until=1, seed=7)
model = AdaptedCLIP(clip, text_adapt_until=1, image_adapt_until=2, levels=[2, 3], relu=False)
model.text_adapter.load_state_dict(ta, strict=True)
clip.to(dev).eval(); model.to(dev).eval()
fails = 0
unit = lambda t: t / t.norm(dim=-1, keepdim=True)
for n in (1, 2, 7, 40, 64, 300):
    tok = torch.zeros(n, 77, dtype=torch.int32)
    for i in range(n):
        eot = int(rng.integers(1, 77))
        tok[i, :eot] = torch.from_numpy(rng.integers(1, cfg.vocab_size - 1, eot).astype(np.int32))
        tok[i, eot] = cfg.vocab_size - 1          # the maximum id marks the end of text
    with torch.no_grad():
        a = model.encode_text(tok.to(dev)).cpu()
        p = clip.encode_text(tok.to(dev)).cpu()
    oa = O.adapted_encode_text(tok, sd, ta, cfg.text.heads, text_adapt_until=1, dtype=torch.float64)
    op = O.encode_text(tok, sd, cfg.text.heads, dtype=torch.float64)
    for name, got, ref in (("adapted", a, oa), ("plain", p, op)):
        err = (unit(got.double()) - unit(ref)).abs()
        bad = err > 1e-3 + 1e-2 * unit(ref).abs()
        if bad.any() or not torch.isfinite(got).all():
            fails += 1
            print(f"FAIL {name} n={n}: {int(bad.sum())}/{bad.numel()} outside, max err {err.max().item():.3e}")
        else:
            print(f"ok   {name} n={n} (M={n * 77}): max err {err.max().item():.2e}")
print("FAILURES:", fails)
sys.exit(1 if fails else 0)
