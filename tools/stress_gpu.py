#!/usr/bin/env python3
"""Randomised shape sweep of the C-ABI kernels against fp64 torch on the GPU (a one-off confidence run, not part of
the test suite): large-M GEMMs on the 256-tile kernels with every epilogue, attention over random (B, L, H, causal),
LayerNorm, the pre-processing kernel on random frame sizes.  usage: python tools/stress_gpu.py [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
sys.path.insert(0, REPO)
import numpy as np
import torch
from aaclip_hip import _lib, engine
from aaclip_hip._lib import F16, BF16

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
TDT = {F16: torch.float16, BF16: torch.bfloat16}
fails = 0


def check(name, got, ref, atol, rtol):
    global fails
    got, ref = got.double(), ref.double()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    if bad.any() or not torch.isfinite(got).all():
        fails += 1
        print(f"FAIL {name}: {int(bad.sum())}/{bad.numel()} outside, max err {err.max().item():.3e}")
    return not bad.any()


def gemm(code, epi, A, W, bias, out, act=0, scale_cols=0, scale=1.0):
    M, K = A.shape
    N = W.shape[0]
    _lib.check(lib.aaclip_gemm(code, epi, A.data_ptr(), K, W.data_ptr(), None if bias is None else bias.data_ptr(),
                               out.data_ptr(), out.shape[1], M, N, K, act, scale_cols, scale, st), "gemm")


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


n_gemm = 0
for _ in range(14):
    code = [F16, BF16][int(rng.integers(2))]
    M = int(rng.choice([4096, 4100, 5000, 8191, 12345, 20000]))
    N = int(rng.choice([256, 768, 1024, 3072]))
    K = int(rng.choice([128, 256, 640, 768, 1024, 4096, 192]))
    A = torch.randn(M, K, device=dev).to(TDT[code])
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(TDT[code])
    bias = torch.randn(N, device=dev) * 0.5
    acc = A.double() @ W.double().t()
    et = 2e-3 if code == F16 else 1.5e-2
    out = torch.empty(M, N, dtype=TDT[code], device=dev)
    gemm(code, _lib.EPI_BIAS, A, W, bias, out, scale_cols=64, scale=0.125)
    ref = acc + bias.double()
    ref[:, :64] *= 0.125
    ok = check(f"bias M{M} N{N} K{K} {code}", out, ref, et, et)
    gemm(code, _lib.EPI_BIAS_GELU, A, W, bias, out)
    ok &= check(f"gelu M{M} N{N} K{K} {code}", out, gelu(acc + bias.double()), et, et)
    x0 = torch.randn(M, N, device=dev) * 2
    x = x0.clone()
    gemm(code, _lib.EPI_BIAS_RESID, A, W, bias, x)
    ok &= check(f"resid M{M} N{N} K{K} {code}", x, x0.double() + acc + bias.double(), 3e-5, 1e-5)
    o32 = torch.empty(M, N, device=dev)
    gemm(code, _lib.EPI_ACT_F32, A, W, None, o32, act=1)
    ok &= check(f"leaky M{M} N{N} K{K} {code}", o32, torch.nn.functional.leaky_relu(acc, 0.01), 3e-5, 1e-5)
    n_gemm += 1
    del A, W, acc, out, x, x0, o32
print(f"gemm: {n_gemm} random shapes x 4 epilogues done")

n_attn = 0
for _ in range(12):
    code = [F16, BF16][int(rng.integers(2))]
    B = int(rng.integers(1, 5)); H = int(rng.choice([1, 2, 4, 12, 16])); causal = int(rng.integers(2))
    L = int(rng.choice([1, 7, 63, 64, 65, 77, 200, 511, 512, 513, 777, 1370, 1500]))
    D = H * 64
    qkv = torch.randn(B * L, 3 * D, device=dev)
    qkv[:, :D] *= 0.5
    q16 = qkv.to(TDT[code])
    ctx = torch.empty(B * L, D, dtype=TDT[code], device=dev)
    _lib.check(lib.aaclip_attention(code, q16.data_ptr(), ctx.data_ptr(), B, L, H, causal, st), "attention")
    q, k, v = q16.double().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=dev, dtype=torch.float64).triu_(1)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, D)
    check(f"attention B{B} L{L} H{H} causal{causal} {code}", ctx, ref, 3e-3 if code == F16 else 2e-2, 2e-2)
    n_attn += 1
print(f"attention: {n_attn} random shapes done")

from oracle import preprocess_oracle as P
n_pre = 0
for _ in range(8):
    h, w = int(rng.integers(20, 1500)), int(rng.integers(20, 1500))
    S = int(rng.choice([70, 224, 518]))
    Bn = int(rng.integers(1, 3))
    imgs = rng.integers(0, 256, (Bn, h, w, 3), dtype=np.uint8)
    got = engine.preprocess(torch.from_numpy(imgs).to(dev), S).cpu().numpy()
    for b in range(Bn):
        if not np.array_equal(got[b], P.preprocess(imgs[b], S)):
            fails += 1
            print(f"FAIL preprocess {h}x{w} -> {S}")
    n_pre += 1
print(f"preprocess: {n_pre} random sizes done (bit-exact)")
print("FAILURES:", fails)
sys.exit(1 if fails else 0)
