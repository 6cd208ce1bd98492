#!/usr/bin/env python3
"""Diagnostic: where a tile's cycles go in the persistent GEMM (needs a -DZ_STAMP build of gemm256z.hip).
usage: python tools/gemm_zstamps.py K N EPI   (EPI 0 bias, 1 gelu, 2 resid, 3 act_f32)"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
os.environ.setdefault("AACLIP_LIB", os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so"))   # A/B variants live in the measurement library (make measure)
from aaclip_hip import _lib
lib = _lib.load()
lib.aaclip_debug_gemm_stamps.restype = C.c_int
lib.aaclip_debug_gemm_stamps.argtypes = [C.POINTER(C.c_double), C.c_int]
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M = 64 * 1370
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
bias = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.float16 if epi < 2 else torch.float32)
st = torch.cuda.current_stream().cuda_stream
lib.aaclip_set_gemm_variant(70)
for _ in range(3):
    _lib.check(lib.aaclip_gemm(_lib.F16, epi, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, 0, 0, 1.0, st))
torch.cuda.synchronize()
o = (C.c_double * 8)()
lib.aaclip_debug_gemm_stamps(o, -1)
names = ["bookkeeping+clear", "first K pair", "middle K tiles", "last K pair", "align barrier", "epilogue", "re-stagger"]
tot = sum(o[k] for k in range(7))
print(f"K={K} N={N} epi={epi}: {o[7]:.0f} tiles, {tot:.0f} memtime ticks per tile (100 MHz ticks -> {tot / 100:.1f} us)")
for k in range(7):
    print(f"  {names[k]:20s} {o[k]:9.0f} ticks  {o[k] / 100:7.2f} us  {100 * o[k] / tot:5.1f} %")
