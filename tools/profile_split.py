#!/usr/bin/env python3
"""One split-fp16 (AACLIP_F16X2) GEMM launch shape for rocprofv3 counter passes (program directly after `--`):
   rocprofv3 --pmc ... -- python3 tools/profile_split.py c_proj     (AACLIP_LIB selects an experiment build)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import _lib, engine
name = sys.argv[1] if len(sys.argv) > 1 else "c_proj"
K, N, epi = {"qkv": (1024, 3072, 0), "out_proj": (1024, 1024, 2), "c_fc": (1024, 4096, 1), "c_proj": (4096, 1024, 2)}[name]
lib = _lib.load()
dev = torch.device("cuda:0")
M = 64 * 1370
st = torch.cuda.current_stream(dev).cuda_stream
torch.manual_seed(1)
A = engine.split_rows(torch.randn(M, K))
A = A.to(dev)
W = engine.split_rows(torch.randn(N, K) * K ** -0.5, weight=True).to(dev)
bias = torch.zeros(N, device=dev)
out = torch.zeros(M, N, dtype=torch.float32, device=dev) if epi == 2 else torch.empty(M, 4 * N, dtype=torch.uint8, device=dev)
ldc = N if epi == 2 else 2 * N
for _ in range(5):
    _lib.check(lib.aaclip_gemm(_lib.F16X2, epi, A.data_ptr(), 2 * K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), ldc, M, N, K,
                               0, 0, 1.0, st), "gemm")
torch.cuda.synchronize()
print("done", name)
