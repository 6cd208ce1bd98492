#!/usr/bin/env python3
"""Randomised sweep of the map / head entry points against the CPU oracle: anomaly map (levels, grid, output size,
per-image or shared anchors, both blur settings), train-mode similarity map, tap head with and without det head.
One-off confidence run.  usage: python tools/stress_heads.py [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
sys.path.insert(0, REPO)
import numpy as np
import torch
from aaclip_hip import engine
from aaclip_hip._lib import F16, F32
from oracle import aaclip_oracle as O
import forward_utils as FU

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
fails = 0


def check(tag, got, ref, atol, rtol):
    global fails
    err = (got.double().cpu() - ref.double()).abs()
    bad = err > atol + rtol * ref.double().abs()
    if bad.any() or not torch.isfinite(got).all():
        fails += 1
        print(f"FAIL {tag}: {int(bad.sum())}/{bad.numel()} outside, max err {err.max().item():.3e}")
    else:
        print(f"ok   {tag}: max err {err.max().item():.2e}")


for _ in range(10):
    B = int(rng.integers(1, 6)); g = int(rng.choice([5, 16, 24, 37])); E = int(rng.choice([256, 512, 768]))
    S = int(rng.choice([70, 224, 518, 333])); NL = int(rng.integers(1, 5))
    domain = ["Industrial", "Medical"][int(rng.integers(2))]
    segs = [torch.nn.functional.normalize(torch.randn(B, g * g, E), dim=-1) for _ in range(NL)]
    per_image = bool(rng.integers(2))
    tf = torch.nn.functional.normalize(torch.randn(B, E, 2) if per_image else torch.randn(E, 2), dim=-2)
    got = FU.calculate_anomaly_map([s.to(dev) for s in segs], tf.to(dev), S, domain=domain)
    ref = O.anomaly_map(segs, tf, S, domain)
    check(f"anomaly_map B{B} g{g} E{E} S{S} levels{NL} {domain} per_image{int(per_image)}", got, ref, 2e-3, 1e-4)
    got = FU.calculate_similarity_map(segs[0].to(dev), tf.to(dev), S, test=False)
    ref = O.similarity_map(segs[0], tf, S, test=False)
    check(f"train map  B{B} g{g} E{E} S{S}", got, ref, 1e-4, 1e-4)

print("FAILURES:", fails)
sys.exit(1 if fails else 0)
