#!/usr/bin/env python3
"""Time the fused attention kernel at the tower's shape (B=64, L=1370, H=16, head 64)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
if "--variant" in sys.argv:   # A/B variants live in the measurement library (make measure); default: the product library
    os.environ.setdefault("AACLIP_LIB", os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so"))
from aaclip_hip import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--L", type=int, default=1370)
ap.add_argument("--H", type=int, default=16)
ap.add_argument("--causal", type=int, default=0)
ap.add_argument("--variant", type=int, default=0, help="1 = force the 128-row kernel")
ap.add_argument("--scale", type=float, default=0.5)
ap.add_argument("--split", action="store_true", help="the split-fp16 kernel (AACLIP_F16X2): split16 rows in, split8 rows out")
ap.add_argument("--plain", action="store_true", help="time the plain contract (aaclip_attention: natural-exp scores, a per-score "
                "multiply) instead of the kernel variant the block path runs (aaclip_attention_log2q)")
a = ap.parse_args()
lib = _lib.load()
lib.aaclip_set_gemm_variant(a.variant << 8)
dev = torch.device("cuda:0")
B, L, H = a.batch, a.L, a.H
D = 64 * H
qkv = torch.randn(B * L, 3 * D, device=dev)
qkv[:, :D] *= a.scale
if a.split:
    from aaclip_hip import engine
    qkv = engine.split16_rows(qkv)
    ctx = torch.empty(B * L, 4 * D, device=dev, dtype=torch.uint8)
else:
    qkv = qkv.half()
    ctx = torch.empty(B * L, D, device=dev, dtype=torch.float16)
code = _lib.F16X2 if a.split else _lib.F16
st = torch.cuda.current_stream().cuda_stream
fn = lib.aaclip_attention if a.plain else lib.aaclip_attention_log2q   # same scores either way: random q
def run():
    _lib.check(fn(code, qkv.data_ptr(), ctx.data_ptr(), B, L, H, a.causal, st))
run(); torch.cuda.synchronize()
fl = 4.0 * B * H * L * L * 64 * (0.5 if a.causal else 1.0)
ts = []
for r in range(a.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
ts.sort()
print(f"attention ({'plain' if a.plain else 'log2q, block path'}) B={B} L={L} H={H}: median {ts[len(ts)//2]:.3f} ms -> {fl/ts[len(ts)//2]/1e9:.0f} TF")
