#!/usr/bin/env python3
"""The text side of the path alone, for rocprofv3 (the program goes directly after `--`):
    rocprofv3 --kernel-trace --stats -d <dir> -o r -- python3 tools/profile_text.py fp16x2
All 240 MVTec prompt sentences (15 classes x 16) through forward_utils.get_adapted_text_embedding: ONE batched
AdaptedCLIP.encode_text call (M = 18 480 rows, 12 causal blocks of width 768 with the text adapters) + the anchor means
(reference forward_utils.py:138-192, model/adapter.py:273-304).  One warm-up call, then `iters` calls."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth
from model.clip import create_model
from model.adapter import AdaptedCLIP
import forward_utils as FU

precision = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=precision, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
model = AdaptedCLIP(clip, relu=False)
model.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
model = model.to(dev).eval()
with torch.no_grad():
    FU.get_adapted_text_embedding(model, "MVTec", dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        anchors = FU.get_adapted_text_embedding(model, "MVTec", dev)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t) / iters
print(f"{precision}: {len(anchors)} classes, 240 sentences per call, {dt * 1e3:.2f} ms per call, "
      f"{240 * 13.57 / dt / 1e3:.0f} TFLOP/s (13.57 GFLOP per sentence)")
