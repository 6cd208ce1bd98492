#!/usr/bin/env python3
"""Writes the B = 4 tower outputs (pooled, tap 6/24) of the loaded library to an .npz: python tools/dump_taps.py out.npz [precision]
(used to compare two builds of the library output for output: AACLIP_LIB selects the build)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import numpy as np, torch
from aaclip_hip import synth
from model.clip import create_model
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=prec, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
clip = clip.to(dev).eval()
g = torch.Generator(device=dev).manual_seed(7)
x = torch.randn(4, 3, 518, 518, generator=g, device=dev)
with torch.no_grad():
    pooled, taps = clip.encode_image(x, [6, 12, 18, 24])
np.savez(sys.argv[1], pooled=pooled.cpu().numpy(), tap6=taps[0].cpu().numpy(), tap24=taps[3].cpu().numpy())
