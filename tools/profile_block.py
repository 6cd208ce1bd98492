#!/usr/bin/env python3
"""Runs a few blocks of the REAL tower path (aaclip_blocks_taps through the Python mirror: LayerNorm folds, the 16-bit-copy
+ row-sum epilogue of out_proj / c_proj, the log2-q attention variant -- what bench.py times) for rocprofv3 counter
passes; the program goes directly after `--`:
    rocprofv3 --pmc ... -- python3 tools/profile_block.py --precision fp16 --blocks 3 --iters 3
Counters are then summarised PER KERNEL INSTANTIATION by tools/rocpd_summary.py pmcjson with a kernel-name substring
(mangled, e.g. 'gemm16_256x_kernelIDF16_Li1ELi0E' = c_fc in fp16; '...Li1ELi4E' = c_fc in fp16x2)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth, engine
from model.clip import create_model

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16x2")
ap.add_argument("--clip-weights", default="fp32", choices=["fp32", "fp16"])
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--blocks", type=int, default=3)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=a.precision, force_image_size=518)
sd = synth.synth_clip_state_dict(cfg, 111)
if a.clip_weights == "fp16":
    sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
clip.load_state_dict(sd, strict=True)
clip = clip.to(dev).eval()
code = engine.dtype_code(a.precision)
B, L, D = a.batch, 1370, 1024
torch.manual_seed(1)
x0 = torch.randn(B * L, D, device=dev)
blocks = list(clip.visual.transformer.resblocks)[: a.blocks]
outs = [torch.empty_like(x0) for _ in blocks]
with torch.no_grad():
    for _ in range(a.iters):
        engine.run_blocks(x0, blocks, B, L, 16, code, x_outs=outs)
torch.cuda.synchronize()
print("done", a.precision, a.blocks, a.iters)
