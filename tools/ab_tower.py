#!/usr/bin/env python3
"""A/B two library settings on the B = 64 tower in ONE process (interleaved rounds): aaclip_set_gemm_variant words,
e.g. 0 vs 131072 (bit 17: no LayerNorm folding).  Prints images/s per setting and whether the pooled outputs are bit-identical.
usage: python tools/ab_tower.py [--variants 0,131072] [--batch 64] [--rounds 3] [--steps 5]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import _lib, synth
from model.clip import create_model

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,131072")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=a.precision, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
clip = clip.to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(5)
images = torch.randn(a.batch, 3, 518, 518, generator=gen, device=dev)
variants = [int(v) for v in a.variants.split(",")]
outs, rates = {}, {v: [] for v in variants}
with torch.no_grad():
    for v in variants:
        assert lib.aaclip_set_gemm_variant(v) == 0, lib.aaclip_last_error()
        pooled, taps = clip.encode_image(images, [6, 12, 18, 24])
        torch.cuda.synchronize()
        outs[v] = (pooled.clone(), taps[-1].clone())
    for r in range(a.rounds):
        for v in variants:
            lib.aaclip_set_gemm_variant(v)
            clip.encode_image(images, [6, 12, 18, 24]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                clip.encode_image(images, [6, 12, 18, 24])
            torch.cuda.synchronize()
            rates[v].append(a.batch * a.steps / (time.perf_counter() - t0))
lib.aaclip_set_gemm_variant(0)
for v in variants[1:]:
    same = torch.equal(outs[variants[0]][0], outs[v][0]) and torch.equal(outs[variants[0]][1], outs[v][1])
    print(f"setting {variants[0]} vs {v}: outputs bit-identical = {same}")
for v in variants:
    rs = sorted(rates[v])
    print(f"setting {v}: median {rs[len(rs)//2]:.1f} images/s  (rounds: {', '.join(f'{x:.1f}' for x in rates[v])})")
