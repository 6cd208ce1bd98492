#!/usr/bin/env python3
"""AdaptedCLIP.forward WITH the IQM branch, a few times, for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_iqm.py`."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth
from model.clip import create_model
from model.adapter import AdaptedCLIP
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=prec, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
model = AdaptedCLIP(clip, relu=False)
model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
model.load_state_dict(synth.synth_iqm_state_dict(cfg, seed=7), strict=False)
model = model.to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1)
images = torch.randn(64, 3, 518, 518, generator=gen, device=dev)
te = torch.nn.functional.normalize(torch.randn(64, 768, 2, generator=gen, device=dev), dim=1)
iqm = not (len(sys.argv) > 2 and sys.argv[2] == "noiqm")     # "noiqm": the same forward without the side branch
with torch.no_grad():
    for _ in range(3):
        model(images, text_embeddings=te if iqm else None)
torch.cuda.synchronize()
print("done")
