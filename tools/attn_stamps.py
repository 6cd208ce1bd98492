#!/usr/bin/env python3
"""Where a key tile of the long-sequence attention kernel spends its cycles (measurement library built with
`make measure MEASURE_DEFS=-DATTN_STAMP`): s_memtime segment sums of waves 0/1 of every workgroup, per tile."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "aa-clip-iqm_amd")]
os.environ.setdefault("AACLIP_LIB", os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so"))
import torch
from aaclip_hip import _lib
lib = _lib.load()
f = lib.aaclip_measure_attn_stamps
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.aaclip_set_gemm_variant(int(os.environ.get("ATTN_VARIANT", "0")) << 8)   # 3 = 8-wave workgroups
dev = torch.device("cuda:0")
B, L, H = 64, 1370, 16
D = 64 * H
qkv = torch.randn(B * L, 3 * D, device=dev)
qkv[:, :D] *= 0.5
qkv = qkv.half()
ctx = torch.empty(B * L, D, device=dev, dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(lib.aaclip_attention_log2q(_lib.F16, qkv.data_ptr(), ctx.data_ptr(), B, L, H, 0, st))
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
f(out, 1)
for _ in range(5):
    _lib.check(lib.aaclip_attention_log2q(_lib.F16, qkv.data_ptr(), ctx.data_ptr(), B, L, H, 0, st))
torch.cuda.synchronize()
f(out, 0)
if int(os.environ.get("ATTN_VARIANT", "0")) != 6:   # block-pipelined kernel (attn16x2)
    names = ["wait+barrier", "DMA issue", "B1 S0 chains", "B2 S1 || exp S0", "check0 (+rare)", "B3 PV0 || exp S1", "check1 (+rare)", "B4 PV1 + sums"]
else:                                          # unit-pipelined experiment (attn16u): two sub-tiles per tile
    names = ["wait+barrier", "DMA issue", "steps a (x2)", "checks a (x2)", "steps b (x2)", "checks b (x2)", "-", "-"]
tiles = out[8]
tot = sum(out[i] for i in range(8))
for i, n in enumerate(names):
    print(f"{n:20s} {out[i] / tiles:8.0f} cycles per tile per wave  ({100.0 * out[i] / tot:.1f} %)")
print(f"total {tot / tiles:.0f} cycles per tile per wave (2 waves share a SIMD)")
print("DMA instructions one by one (stamped, inside the DMA issue segment):", [round(out[i] / tiles) for i in range(9, 13)])
