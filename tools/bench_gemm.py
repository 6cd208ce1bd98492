#!/usr/bin/env python3
"""A/B the GEMM kernels on the tower's shapes in ONE process (interleaved rounds,
random data).  usage: python tools/bench_gemm.py [--batch 64] [--rounds 5]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
os.environ.setdefault("AACLIP_LIB", os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so"))   # A/B variants live in the measurement library (make measure)
from aaclip_hip import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="1,2")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--only", default="")
ap.add_argument("--m", type=int, default=0, help="override the row count (default batch*1370)")
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
code = {"f16": _lib.F16, "bf16": _lib.BF16}[a.dtype]
tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
M = a.m or a.batch * 1370
shapes = [("qkv", _lib.EPI_BIAS, 3072, 1024), ("out_proj", _lib.EPI_BIAS_RESID, 1024, 1024),
          ("c_fc", _lib.EPI_BIAS_GELU, 4096, 1024), ("c_proj", _lib.EPI_BIAS_RESID, 1024, 4096),
          ("adapter", _lib.EPI_ACT_F32, 1024, 1024), ("seg_proj", _lib.EPI_ACT_F32, 768, 1024),
          ("k4096_f32out", _lib.EPI_ACT_F32, 1024, 4096), ("n4096_f32out", _lib.EPI_ACT_F32, 4096, 1024),
          ("k64_n4096", _lib.EPI_ACT_F32, 4096, 64), ("k128_n4096", _lib.EPI_ACT_F32, 4096, 128),
          ("k256_n4096", _lib.EPI_ACT_F32, 4096, 256)]
if a.only:
    shapes = [s for s in shapes if s[0] in a.only.split(",")]
if any(v in (4, 5, 11, 12, 13, 14, 15, 16, 46) for v in [int(v) for v in a.variants.split(",")]):
    shapes = [s for s in shapes if s[1] == _lib.EPI_ACT_F32]   # timing ablations exist for the fp32-out epilogue only
variants = [int(v) for v in a.variants.split(",")]
st = torch.cuda.current_stream().cuda_stream
res = {}
for name, epi, N, K in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.float32).to(tdt)
    W = (torch.randn(N, K, device=dev, dtype=torch.float32) * K ** -0.5).to(tdt)
    bias = torch.randn(N, device=dev)
    out_t = torch.empty(M, N, device=dev, dtype=tdt)
    out_f = torch.zeros(M, N, device=dev, dtype=torch.float32)
    out = out_t if epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU) else out_f
    def run(v):
        lib.aaclip_set_gemm_variant(v)
        _lib.check(lib.aaclip_gemm(code, epi, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K,
                                   1, 0, 1.0, st))
    outs = {}
    for v in variants:
        out_f.zero_()
        run(v); torch.cuda.synchronize()
        outs[v] = out.float().clone()
    if len(variants) > 1:
        d = (outs[variants[0]] - outs[variants[1]]).abs().max().item()
        print(f"{name}: max |v{variants[0]} - v{variants[1]}| = {d:.3e}")
    times = {v: [] for v in variants}
    for r in range(a.rounds):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(v)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * M * N * K
    for v in variants:
        t = sorted(times[v])
        print(f"{name:9s} v{v}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms  -> {fl / t[len(t)//2] / 1e9:.0f} TF (median) {fl / t[0] / 1e9:.0f} TF (best)")
    del A, W, out_t, out_f
lib.aaclip_set_gemm_variant(0)
