#!/usr/bin/env python3
"""Cost of the IQM side branch: AdaptedCLIP.forward with and without text_embeddings, plus the fused
IQM/text map, at the benchmark batch.  usage: python tools/bench_iqm.py [--batch 64] [--steps 3]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth
from model.clip import create_model
from model.adapter import AdaptedCLIP

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=a.precision, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
model = AdaptedCLIP(clip, relu=False)
model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
model.load_state_dict(synth.synth_iqm_state_dict(cfg, seed=7), strict=False)
model = model.to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1)
B = a.batch
images = torch.randn(B, 3, 518, 518, generator=gen, device=dev)
te = torch.nn.functional.normalize(torch.randn(B, 768, 2, generator=gen, device=dev), dim=1)

def timed(fn):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.steps

t_plain = timed(lambda: model(images))
t_iqm = timed(lambda: model(images, text_embeddings=te))
print(f"B={B} {a.precision}: forward {t_plain:.2f} ms, with the IQM branch {t_iqm:.2f} ms (+{t_iqm - t_plain:.2f} ms, "
      f"+{100 * (t_iqm / t_plain - 1):.1f} %)")
