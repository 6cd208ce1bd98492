#!/usr/bin/env python3
"""Soak: the same batch through the tower N times; every output of EVERY pass is compared bit for bit with the first pass (a race, a stale
workspace or an occasional fault would show as a mismatch or non-finite value).  usage: python tools/soak.py [steps] [precision]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth
from model.clip import create_model
from model.adapter import AdaptedCLIP
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=prec, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
model = AdaptedCLIP(clip, relu=False)
model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
model.load_state_dict(synth.synth_iqm_state_dict(cfg, seed=7), strict=False)
model = model.to(dev).eval()
img = torch.randn(64, 3, 518, 518, device=dev)
te = torch.nn.functional.normalize(torch.randn(64, 768, 2, device=dev), dim=1)
bad = compared = 0
t0 = time.time()
with torch.no_grad():
    ref = None
    for i in range(steps):
        seg, det, iq = model(img, text_embeddings=te)
        cur = list(seg) + [det, iq.last_hidden_state]
        if ref is None:
            ref = [t.clone() for t in cur]
            assert all(torch.isfinite(t).all() for t in ref)
        else:
            # EVERY pass is compared, on the device (torch.equal per output: < 1 ms beside a > 100 ms pass)
            same = all(torch.equal(a, b) for a, b in zip(cur, ref))
            compared += 1
            if not same:
                bad += 1
                print(f"step {i}: outputs differ from the first pass", flush=True)
        if i % 25 == 0 or i == steps - 1:
            print(f"step {i}: {time.time() - t0:.0f} s", flush=True)
print(f"{prec}: {steps} passes of the full path + IQM branch, {compared} compared with the first, mismatches: {bad}")
sys.exit(1 if bad else 0)
