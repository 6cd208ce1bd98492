#!/usr/bin/env python3
"""Experiment: one batch of 64 on one stream vs two half batches on two streams."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth
from model.clip import create_model
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision="fp16", force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
clip.to(dev).eval()
img = torch.randn(64, 3, 518, 518, device=dev)
def run_one():
    clip.encode_image(img, [6, 12, 18, 24])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        clip.encode_image(img[:32], [6, 12, 18, 24])
    with torch.cuda.stream(s2):
        clip.encode_image(img[32:], [6, 12, 18, 24])
    cur.wait_stream(s1); cur.wait_stream(s2)
s3, s4 = torch.cuda.Stream(), torch.cuda.Stream()
def run_four():
    cur = torch.cuda.current_stream()
    ss = (s1, s2, s3, s4)
    for st in ss: st.wait_stream(cur)
    for i, st in enumerate(ss):
        with torch.cuda.stream(st):
            clip.encode_image(img[16 * i:16 * i + 16], [6, 12, 18, 24])
    for st in ss: cur.wait_stream(st)
def run_seq_halves():
    clip.encode_image(img[:32], [6, 12, 18, 24]); clip.encode_image(img[32:], [6, 12, 18, 24])
with torch.no_grad():
    for name, fn in (("one stream, B=64", run_one), ("two halves sequential", run_seq_halves), ("two halves, two streams", run_two), ("four quarters, four streams", run_four)):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(f"{name}: {dt*1e3:.1f} ms/step -> {64/dt:.0f} img/s")
