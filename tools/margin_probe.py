#!/usr/bin/env python3
"""How much margin does fp16x2 keep to the north-star tolerance on inputs OTHER than the golden record?

For several seeds: fresh synthetic CLIP + adapter weights, fresh images (B = 4, full size), the full AA-CLIP visual side in
the exact-fp32 mode (0.006 of the bound against the reference on the golden record, so it stands in for the reference
here) and in fp16x2; worst |err| / (1e-3 + 1e-2 |ref|) over the raw taps, the unit seg tokens, the det token and the
per-level pre-blur maps.  usage: python tools/margin_probe.py [seeds=4] [precision=fp16x2]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "aa-clip-iqm_amd"), REPO]
import torch
from aaclip_hip import engine, synth
from model.clip import create_model
from model.adapter import AdaptedCLIP
import forward_utils as FU

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()


def build(precision, seed, exact16):
    clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=precision, force_image_size=518)
    sd = synth.synth_clip_state_dict(cfg, seed)
    if exact16:
        sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
    clip.load_state_dict(sd, strict=True)
    model = AdaptedCLIP(clip, relu=False)
    model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=seed), strict=True)
    return clip, model.to(dev).eval()


def ratio(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).abs() / (1e-3 + 1e-2 * b.abs())).max())


worst_all = 0.0
for i in range(nseeds):
    seed = 1000 + 17 * i
    exact16 = bool(i & 1)
    img = synth.synth_images(4, 518, seed=seed).to(dev)
    anchors = torch.nn.functional.normalize(torch.randn(768, 2, generator=torch.Generator().manual_seed(seed)), dim=0).to(dev)
    outs = {}
    for p in ("fp32", prec):
        clip, model = build(p, seed, exact16)
        with torch.no_grad():
            pooled, taps = clip.encode_image(img, [6, 12, 18, 24])
            seg, det, _ = model(img)
            maps = [FU.calculate_similarity_map(s, anchors, 37)[:, 1] for s in seg]      # grid size: the pre-blur map of a level
        outs[p] = {"pooled": pooled.float().cpu(), "taps": [t.float().cpu() for t in taps], "seg": [s.float().cpu() for s in seg],
                   "det": det.float().cpu(), "maps": [m.float().cpu() for m in maps]}
        del clip, model
        torch.cuda.empty_cache()
    r, t = outs["fp32"], outs[prec]
    rows = {"pooled": ratio(t["pooled"], r["pooled"]), "det": ratio(t["det"], r["det"])}
    for k in range(4):
        rows[f"tap{6 * (k + 1)}"] = ratio(t["taps"][k], r["taps"][k])
        rows[f"seg{k}"] = ratio(t["seg"][k], r["seg"][k])
    for k, (a, b) in enumerate(zip(t["maps"], r["maps"])):
        rows[f"map{k}"] = ratio(a, b)
    worst = max(rows.values())
    worst_all = max(worst_all, worst)
    print(f"seed {seed} ({'fp16-exact' if exact16 else 'fp32'} CLIP weights): worst ratio {worst:.3f} at {max(rows, key=rows.get)};  "
          + "  ".join(f"{k} {v:.3f}" for k, v in rows.items()), flush=True)
print(f"{prec}: worst ratio to the north-star bound over {nseeds} seeds: {worst_all:.3f} ({'inside' if worst_all <= 1 else 'OUTSIDE'})")
