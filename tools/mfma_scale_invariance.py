#!/usr/bin/env python3
"""Is a softmax-weighted sum on the fp16 MFMAs invariant to the reference point of the probabilities?
P = 2^(s - ref) is rounded to fp16 and multiplied with fp16 V on the matrix cores (fp32 accumulation, the library's
GEMM with fp32 output); ref = row maximum - lag.  Mathematically sum(P V) / sum(P) does not depend on lag.  Measured
against fp64 for several lags (the split attention kernel's context error grew with a lagging reference point)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import engine
from aaclip_hip._lib import F16, EPI_ACT_F32
dev = torch.device("cuda:0")
torch.manual_seed(0)
rows, keys, cols = 4096, 1344, 128
s = torch.randn(rows, keys, device=dev, dtype=torch.float64) * 3.0
v = torch.randn(keys, cols, device=dev, dtype=torch.float64).half()
vt = v.t().contiguous()
mx = s.max(1, keepdim=True).values
pref = torch.exp2(s - mx)
ref = (pref @ v.double()) / pref.sum(1, keepdim=True)
for lag in (0, 1, 4, 8, 12):
    p32 = torch.exp2((s - (mx - lag)).float())
    p16 = p32.half()
    num = torch.empty(rows, cols, dtype=torch.float32, device=dev)
    engine.gemm(F16, EPI_ACT_F32, p16.contiguous(), vt, None, num)
    out = num.double() / p32.double().sum(1, keepdim=True)
    exact16 = (p16.double() @ v.double()) / p32.double().sum(1, keepdim=True)      # the same fp16 P, summed in fp64
    e = (out - ref).pow(2).mean().sqrt().item()
    e2 = (exact16 - ref).pow(2).mean().sqrt().item()
    e3 = (out - exact16).pow(2).mean().sqrt().item()
    print(f"lag {lag:2d}: rms error vs fp64 {e:.3e}; of which fp16 rounding of P (summed in fp64) {e2:.3e}; MFMA accumulation alone {e3:.3e}")
