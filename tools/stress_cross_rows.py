#!/usr/bin/env python3
"""Randomised sweep of aaclip_cross_rows_levels (csrc/iqm.hip, cross_rows_mfma_kernel) against fp64: batch sizes on both
sides of the slicing thresholds, 1-4 segments, 1-16 effective queries, key counts around the 32-key tile and the slice
boundaries, leading rows skipped, fp16 / bf16 / split8 strides, peaked and flat score distributions.
usage: python tools/stress_cross_rows.py [seed] [cases]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import numpy as np
import torch
from aaclip_hip import engine

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
fails = 0
for c in range(cases):
    B = int(rng.choice([1, 2, 3, 7, 16, 64, 130]))
    R = int(rng.integers(1, 17))
    nseg = int(rng.integers(1, 5))
    Dk = int(rng.choice([768, 1024]))
    Lk = int(rng.choice([1, 5, 31, 32, 33, 63, 64, 65, 97, 300, 1369, 2000]))
    if B * nseg * Lk * Dk > 3.0e8:
        Lk = min(Lk, 300)
    row0 = int(rng.integers(0, 3))
    rpi = row0 + Lk + int(rng.integers(0, 3))
    fmt = str(rng.choice(["fp16", "bf16", "split8"]))
    sharp = float(rng.choice([0.3, 1.0, 4.0]))
    tdt = torch.bfloat16 if fmt == "bf16" else torch.float16
    qt = torch.randn(B * R, nseg * Dk, device=dev) * sharp * Dk ** -0.5
    xs = [(torch.randn(B * rpi, Dk, device=dev) * (0.5 + s) + 0.2 * s).to(tdt) for s in range(nseg)]
    if fmt == "split8":
        levels = []
        for x in xs:
            rec = torch.full((B * rpi, 4 * Dk), 0x7F, dtype=torch.uint8, device=dev)
            rec[:, :2 * Dk] = x.contiguous().view(torch.uint8).view(B * rpi, 2 * Dk)
            levels.append(rec)
    else:
        levels = xs
    out = engine.cross_rows_levels(qt, levels, B, R, rpi, row0, Lk, Dk)
    q64 = qt.double().view(B, R, nseg, Dk)
    keys = [x.double().view(B, rpi, Dk)[:, row0:row0 + Lk] for x in xs]
    sc = torch.cat([torch.einsum("brd,bjd->brj", q64[:, :, s], keys[s]) for s in range(nseg)], -1)
    p = torch.softmax(sc, -1)
    ref = torch.stack([torch.einsum("brj,bjd->brd", p[:, :, s * Lk:(s + 1) * Lk], keys[s]) for s in range(nseg)], 2)
    err = (out.double().view_as(ref) - ref).abs().max().item()
    tol = (6e-4 if fmt != "bf16" else 5e-3) * max(ref.abs().max().item(), 1e-6)
    tag = f"B{B} R{R} seg{nseg} Dk{Dk} Lk{Lk} row0 {row0} rpi{rpi} {fmt} sharp{sharp}"
    if not torch.isfinite(out).all() or err > tol:
        fails += 1
        print(f"FAIL {tag}: max err {err:.3e} (tol {tol:.3e})")
    else:
        print(f"ok   {tag}: max err {err:.2e}")
    del qt, xs, levels, out, ref, p, sc, keys
print("FAILURES:", fails)
sys.exit(1 if fails else 0)
