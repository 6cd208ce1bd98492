#!/usr/bin/env python3
"""A/B attention kernel variants of the measurement library in ONE process (interleaved rounds, same data), with the
maximum difference of their outputs.  usage: python tools/attn_ab.py --variants 0,6 [--L 1370] [--batch 64] [--data zeros]
Variants: 0 product kernel (attn16x2), 1 128-query kernel, 2 attn16p, 3 8-wave attn16x2, 6 unit-pipelined experiment (attn16u)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
os.environ.setdefault("AACLIP_LIB", os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so"))
from aaclip_hip import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--L", type=int, default=1370)
ap.add_argument("--H", type=int, default=16)
ap.add_argument("--variants", default="0,6")
ap.add_argument("--scale", type=float, default=0.5)
ap.add_argument("--data", default="randn", choices=["randn", "zeros", "const"], help="operand values: the chip is power-limited, so the bits that toggle set the clock")
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
B, L, H = a.batch, a.L, a.H
D = 64 * H
torch.manual_seed(0)
qkv = torch.randn(B * L, 3 * D, device=dev)
qkv[:, :D] *= a.scale
if a.data == "zeros":
    qkv.zero_()
elif a.data == "const":
    qkv.fill_(0.25)
qkv = qkv.half()
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in a.variants.split(",")]
outs, times = {}, {v: [] for v in variants}
def run(v, ctx):
    if lib.aaclip_set_gemm_variant(v << 8) != 0:
        raise SystemExit(f"variant {v} refused: {lib.aaclip_last_error().decode()}")
    _lib.check(lib.aaclip_attention_log2q(_lib.F16, qkv.data_ptr(), ctx.data_ptr(), B, L, H, 0, st))
for v in variants:
    outs[v] = torch.zeros(B * L, D, device=dev, dtype=torch.float16)
    run(v, outs[v])
torch.cuda.synchronize()
for v in variants[1:]:
    print(f"max |v{variants[0]} - v{v}| = {(outs[variants[0]].float() - outs[v].float()).abs().max().item():.3e}")
for r in range(a.rounds):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(v, outs[v])
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 5)
fl = 4.0 * B * H * L * L * 64
for v in variants:
    t = sorted(times[v])
    print(f"variant {v}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f}  -> {fl / t[len(t)//2] / 1e9:.0f} TF")
lib.aaclip_set_gemm_variant(0)
