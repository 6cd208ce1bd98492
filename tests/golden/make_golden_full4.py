#!/usr/bin/env python3
"""Full-size B = 4 golden record from the REFERENCE itself (build container only; stubs as in make_golden.py).

Why B = 4: M = 4*1370 = 5480 rows >= 4096 puts the build on its large-batch kernels (256x256-tile GEMMs with the
LayerNorm passes folded into the QKV / c_fc products) -- the regime bench.py measures at B = 64.  full.npz (B = 2)
only reaches the small-batch kernels.  Recorded (sampled values + checksums, as in make_golden.py):
  * AdaptedCLIP.forward(img4): seg tokens of the 4 levels, det token, the residual stream at the 4 taps
  * the pre-blur test-mode map of every level on the 'bottle' anchors of full.npz, and their level sum
  * CLIP.encode_image(img4, [6, 12, 18, 24]) (BASELINE.json config 2 as written): pooled + the 4 raw taps

Usage:  python tests/golden/make_golden_full4.py        (~1 min of reference time on 8 threads)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

synth = MG.synth


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    A, C, M, TK, FU, CONST = MG._stub_and_import_reference()
    cfg = synth.ClipCfg()
    clip = C.create_model("ViT-L-14-336", img_size=518, pretrained=None, force_image_size=518)
    clip.load_state_dict(synth.synth_clip_state_dict(cfg, seed=111), strict=True)
    model = A.AdaptedCLIP(clip, relu=False).eval()
    model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
    model.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
    anchors = torch.from_numpy(np.load(os.path.join(HERE, "full.npz"))["full.anchors_bottle"])

    B = 4
    img = synth.synth_images(B, 518, seed=4111)
    out = {"full4.seed": np.int64(4111)}
    stream = []
    hook = model.image_encoder.ln_post.register_forward_pre_hook(lambda m, inp: stream.append(inp[0].detach().clone()))
    with torch.no_grad():
        seg, det, iq = model(img)
    hook.remove()
    assert iq is None and len(stream) >= 4
    for i, s_ in enumerate(seg):
        MG.put(out, f"full4.seg{i}", s_)
    out["full4.det"] = det.numpy()
    for i, s_ in enumerate(stream[:4]):
        MG.put(out, f"full4.stream{i}", s_)
    tfb = anchors.unsqueeze(0).repeat(B, 1, 1)
    total = 0
    with torch.no_grad():
        for i, s_ in enumerate(seg):
            sc = 100.0 * torch.matmul(s_, tfb)
            pp = sc.permute(0, 2, 1).view(B, 2, 37, 37)
            pre = (pp[:, 1] + 1 - pp[:, 0]) / 2
            out[f"full4.map_pre_blur{i}"] = pre.numpy()
            total = total + pre
        out["full4.map_pre_blur_sum"] = total.numpy()
        # the image-level score as the reference really evaluates it (test_last.py:90-91 broadcast)
        pred = det @ tfb
        out["full4.image_pred_quirk"] = ((pred[:, 1] + 1) / 2).numpy()
        pooled, taps = clip.encode_image(img, [6, 12, 18, 24])
    out["full4.pooled"] = pooled.numpy()
    for k, t in zip((6, 12, 18, 24), taps):
        MG.put(out, f"full4.tap{k}", t)
    np.savez_compressed(os.path.join(HERE, "full4.npz"), **out)
    print("full4 fixtures written:", sorted(k for k in out if k.endswith(".shape")))


if __name__ == "__main__":
    main()
