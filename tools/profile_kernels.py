#!/usr/bin/env python3
"""Launch ONE product kernel of the B = 64 tower a few times, for rocprofv3 counter passes (program goes directly
after `--`):   rocprofv3 --pmc SQ_WAVE_CYCLES ... -d gpurun_out/pmc_x -- python3 tools/profile_kernels.py --kernel attn
Kernels: attn (fused attention, L = 1370, H = 16), qkv / out_proj / c_fc / c_proj (the block's four GEMMs with their
real epilogues).  Random data (clocks and counters on zeros mislead, MI355X_MICROARCH.md)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", required=True, choices=["attn", "qkv", "out_proj", "c_fc", "c_proj"])
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
code = {"f16": _lib.F16, "bf16": _lib.BF16}[a.dtype]
tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
st = torch.cuda.current_stream().cuda_stream
B, L, H, D = a.batch, 1370, 16, 1024
M = B * L
torch.manual_seed(1)
if a.kernel == "attn":
    qkv = torch.randn(M, 3 * D, device=dev, dtype=torch.float32)
    qkv[:, :D] *= 0.125 * 1.4426950408889634
    qkv = qkv.to(tdt)
    ctx = torch.empty(M, D, device=dev, dtype=tdt)
    # the kernel variant the block path runs: q arrives in log2 units (the plain aaclip_attention contract applies the
    # factor per score in fp32: 64 more VALU instructions per tile)
    run = lambda: _lib.check(lib.aaclip_attention_log2q(code, qkv.data_ptr(), ctx.data_ptr(), B, L, H, 0, st))
else:
    epi, N, K = {"qkv": (_lib.EPI_BIAS, 3072, 1024), "out_proj": (_lib.EPI_BIAS_RESID, 1024, 1024),
                 "c_fc": (_lib.EPI_BIAS_GELU, 4096, 1024), "c_proj": (_lib.EPI_BIAS_RESID, 1024, 4096)}[a.kernel]
    A = torch.randn(M, K, device=dev, dtype=torch.float32).to(tdt)
    W = (torch.randn(N, K, device=dev, dtype=torch.float32) * K ** -0.5).to(tdt)
    bias = torch.randn(N, device=dev)
    out = (torch.empty(M, N, device=dev, dtype=tdt) if epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU)
           else torch.zeros(M, N, device=dev, dtype=torch.float32))
    run = lambda: _lib.check(lib.aaclip_gemm(code, epi, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                             N, M, N, K, 0, 0, 1.0, st))
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
print("done", a.kernel, a.iters)
