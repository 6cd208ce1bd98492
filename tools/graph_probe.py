#!/usr/bin/env python3
"""Does replaying the B = 64 tower step as a captured HIP graph beat launching its ~230 kernels one by one?
(one process, interleaved rounds)  usage: python tools/graph_probe.py [--batch 64]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import synth
from model.clip import create_model
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=a.precision, force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
clip = clip.to(dev).eval()
images = torch.randn(a.batch, 3, 518, 518, generator=torch.Generator(device=dev).manual_seed(5), device=dev)
with torch.no_grad():
    for _ in range(2):
        pooled, taps = clip.encode_image(images, [6, 12, 18, 24])
    torch.cuda.synchronize()
    ref = pooled.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        clip.encode_image(images, [6, 12, 18, 24])
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        gp, gt = clip.encode_image(images, [6, 12, 18, 24])
    g.replay(); torch.cuda.synchronize()
    print("graph replay output bit-identical to eager:", torch.equal(gp, ref))
    rates = {"eager": [], "graph": []}
    for r in range(a.rounds):
        for mode in ("eager", "graph"):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                if mode == "eager":
                    clip.encode_image(images, [6, 12, 18, 24])
                else:
                    g.replay()
            torch.cuda.synchronize()
            rates[mode].append(a.batch * a.steps / (time.perf_counter() - t0))
for m, v in rates.items():
    print(f"{m}: " + ", ".join(f"{x:.1f}" for x in v) + " images/s")
