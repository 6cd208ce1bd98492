#!/usr/bin/env python3
"""Randomised sweep of the attention kernels against fp64 torch: the long-sequence 16-bit kernel (L >= 512) and the
fp32 kernels (MFMA for L >= 64, VALU below; any L), random (B, L, H, causal, dtype, score scale), with score outliers
planted at random tiles so that every re-base path runs (a one-off confidence run, not part of the test suite).
usage: python tools/stress_attention.py [seed] [cases]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import numpy as np
import torch
from aaclip_hip import _lib
from aaclip_hip._lib import F16, BF16, F32

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
TDT = {F16: torch.float16, BF16: torch.bfloat16, F32: torch.float32}
fails = 0
for c in range(cases):
    code = [F16, BF16, F32][int(rng.integers(3))]
    B = int(rng.integers(1, 4)); H = int(rng.choice([1, 2, 3, 16])); causal = int(rng.integers(2))
    L = int(rng.choice([512, 513, 575, 576, 577, 640, 767, 1024, 1025, 1370, 1535, 1536, 1537, 2000]))
    if code == F32:   # the fp32 kernels serve every length (text tower 77, V-V attention over the batch axis, ...)
        L = int(rng.choice([1, 2, 31, 33, 63, 64, 65, 77, 96, 127, 128, 129, 160, 255, 257, 513, 1370]))
    D = H * 64
    scale = float(rng.choice([0.05, 0.3, 0.6, 1.0]))
    qkv = torch.randn(B * L, 3 * D, device=dev)
    qkv[:, :D] *= scale
    n_out = int(rng.integers(0, 6))
    for _ in range(n_out):      # outlier keys: a large component along a direction many queries share
        b, h, key = int(rng.integers(B)), int(rng.integers(H)), int(rng.integers(L))
        d = int(rng.integers(64))
        qkv[b * L:(b + 1) * L, h * 64 + d] += float(rng.choice([0.5, 1.0, 2.0]))
        qkv[b * L + key, D + h * 64 + d] += float(rng.choice([8.0, 30.0, 90.0, -40.0]))
    log2q = int(rng.integers(2)) if code != F32 else 0   # 1: the block path's variant, q in log2 units
    q16 = qkv.to(TDT[code])
    ctx = torch.full((B * L, D), float("nan"), dtype=TDT[code], device=dev)
    fn = lib.aaclip_attention_log2q if log2q else lib.aaclip_attention
    _lib.check(fn(code, q16.data_ptr(), ctx.data_ptr(), B, L, H, causal, st), "attention")
    q, k, v = q16.double().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2)
    if log2q:
        s = s * 0.6931471805599453
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=dev, dtype=torch.float64).triu_(1)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, D)
    got = ctx.double()
    err = (got - ref).abs()
    atol, rtol = {F16: (3e-3, 1e-2), BF16: (2.5e-2, 3e-2), F32: (2e-5, 1e-4)}[code]
    bad = err > atol + rtol * ref.abs()
    ok = bool(torch.isfinite(got).all()) and not bool(bad.any())
    if not ok:
        fails += 1
        print(f"FAIL B{B} L{L} H{H} causal{causal} dtype{code} log2q{log2q} scale{scale} outliers{n_out}: {int(bad.sum())} outside, "
              f"max err {err.max().item():.3e}, finite {bool(torch.isfinite(got).all())}")
print(f"{cases} cases, FAILURES: {fails}")
sys.exit(1 if fails else 0)
