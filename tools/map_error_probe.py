#!/usr/bin/env python3
"""Where does the anomaly-map error of the 16-bit path come from?  

Full-size AdaptedCLIP, B = 4 (large-batch kernels), against the REFERENCE's own numbers (tests/golden/full4.npz):
residual stream at the taps, unit seg tokens, pre-blur maps per level and their sum -- for
  fp16            everything on the fp16 MFMA path (the product default)
  fp16+f32head    fp16 tower, tap heads (ln_post -> seg_proj -> normalise) on the exact-fp32 path
  fp16+f64head    fp16 tower, tap heads recomputed on the CPU in fp64 from the GPU's tap stream (tower error only)
  fp32            everything on the exact-fp32 MFMA path
Prints max |err|, max err / (1e-3 + 1e-2 |ref|) (north-star tolerance: must be <= 1) and the violating fraction.
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "aa-clip-iqm_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np
import torch
from aaclip_hip import engine, synth
from aaclip_hip._lib import F16, F32

G = np.load(os.path.join(REPO, "tests", "golden", "full4.npz"))
dev = torch.device("cuda:0")
T = torch.from_numpy


def stats(name, a, ref):
    a, ref = a.double().cpu().reshape(-1), ref.double().reshape(-1)
    err = (a - ref).abs()
    ratio = err / (1e-3 + 1e-2 * ref.abs())
    print(f"  {name:28s} max|err| {err.max():.3e}  rms {err.pow(2).mean().sqrt():.3e}  max ratio {ratio.max():.2f}  "
          f"viol {float((ratio > 1).double().mean()) * 100:.3f} %   (|ref| max {ref.abs().max():.2f})")


def sampled(name, t):
    f = t.detach().reshape(-1).cpu()
    return f[T(G[f"{name}.idx"])], T(G[f"{name}.val"])


def build(precision):
    from model.clip import create_model
    from model.adapter import AdaptedCLIP
    cfg = synth.ClipCfg()
    clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=precision, force_image_size=518)
    clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
    m = AdaptedCLIP(clip, relu=False)
    m.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
    m.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
    return m.to(dev).eval()


def run(model, code, head_code):
    v = model.image_encoder
    img = synth.synth_images(4, 518, seed=int(G["full4.seed"])).to(dev)
    xs, B, L = engine.patch_embed(img, v, code)
    blocks = list(v.transformer.resblocks)
    ad = model.image_adapter["layer_adapters"]
    taps, segs, run_, aw = [], [], [], []
    for i, blk in enumerate(blocks):
        run_.append(blk); aw.append(ad[i].weight if i < 6 else None)
        if (i + 1) in (6, 12, 18, 24):
            engine.run_blocks(xs, run_, B, L, v.num_heads, code, adapter_weights=aw, mix=0.1)
            run_, aw = [], []
            taps.append(xs.clone())
            k = len(segs)
            if head_code == "f64":
                x = xs.view(B, L, -1)[:, 1:].double().cpu()
                h = torch.nn.functional.layer_norm(x, (1024,), v.ln_post.weight.double().cpu(), v.ln_post.bias.double().cpu(), 1e-5)
                y = h @ model.image_adapter["seg_proj"][k].weight.double().cpu().t()
                segs.append(torch.nn.functional.normalize(y, dim=-1))
            else:
                seg, _ = engine.tap_head(xs, v.ln_post, model.image_adapter["seg_proj"][k].weight, False, B, L, head_code)
                segs.append(seg)
    return taps, segs


def report(tag, taps, segs):
    print(tag)
    anchors = T(np.load(os.path.join(REPO, "tests", "golden", "full.npz"))["full.anchors_bottle"]).double()
    total = 0
    for i in range(4):
        a, r = sampled(f"full4.stream{i}", taps[i].view(4, 1370, 1024)[:, 1:].permute(1, 0, 2))
        stats(f"stream{i} (tap, sampled)", a, r)
    for i in range(4):
        a, r = sampled(f"full4.seg{i}", segs[i])
        stats(f"seg{i} (unit rows, sampled)", a, r)
    for i in range(4):
        s = 100.0 * (segs[i].double().cpu() @ anchors)
        pre = ((s[..., 1] + 1 - s[..., 0]) / 2).view(4, 37, 37)
        stats(f"pre-blur map {i}", pre, T(G[f"full4.map_pre_blur{i}"]))
        total = total + pre
    stats("pre-blur map, level sum", total, T(G["full4.map_pre_blur_sum"]))


with torch.no_grad():
    m16 = build("fp16")
    report("== fp16 everything", *run(m16, F16, F16))
    report("== fp16 tower + exact-fp32 heads", *run(m16, F16, F32))
    report("== fp16 tower + fp64 CPU heads (tower error only)", *run(m16, F16, "f64"))
    del m16
    m32 = build("fp32")
    report("== fp32 everything", *run(m32, F32, F32))
