#!/usr/bin/env python3
"""Diagnostic: per-segment cycle shares of the staggered GEMM K loop (stamp build, variant 17)."""
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
import sys as _s
K = int(_s.argv[1]) if len(_s.argv) > 1 else 4096
N = int(_s.argv[2]) if len(_s.argv) > 2 else 1024
M = 64 * 1370
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
out = torch.zeros(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for v in (17,):
    lib.aaclip_set_gemm_variant(v)
    for _ in range(3):
        _lib.check(lib.aaclip_gemm(_lib.F16, _lib.EPI_ACT_F32, A.data_ptr(), K, W.data_ptr(), None, out.data_ptr(), N, M, N, K, 0, 0, 1.0, st))
    torch.cuda.synchronize()
    o = (C.c_double * 6)()
    lib.aaclip_debug_gemm_stamps(o, 16384)
    nk = K // 64
    print(f"entry -> address setup done (before first DMA): {o[5]:.0f} cycles")
    print(f"per tile per wave (cycles): entry->K loop {o[3]:.0f}   K loop {(o[0]+o[1]+o[2]):.0f}   epilogue incl. store drain {o[4]:.0f}")
    print(f"per K tile per wave (cycles): load+barrier {o[0]/nk:.0f}  compute {o[1]/nk:.0f}  barrier-after-compute {o[2]/nk:.0f}  total {(o[0]+o[1]+o[2])/nk:.0f}  (4 phases per K tile; MFMA own time 4 x 256 = 1024)")
