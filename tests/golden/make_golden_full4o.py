#!/usr/bin/env python3
"""Full-size B = 4 record from the REFERENCE with OUTLIER-CHANNEL weights (build container only).

Every other record is Gaussian-initialised weights: residual-stream values of O(1..4).  Trained ViT-L/14 checkpoints
carry a few channels in the hundreds, LayerNorm gains that compensate for them and hidden units far outside the e4m3
range of the fp16x2 mode's correction planes.  `aaclip_hip.synth.outlier_edit` builds such a model from the seeded
weights (what it changes is listed in its docstring); this script runs the reference on it, exactly as
make_golden_full4.py does on the plain weights, and records the same outputs under `full4o.*` plus the magnitudes the
edit produced (`full4o.stream_absmax`, `full4o.gelu_absmax`) so that the test can assert the record really is an
outlier case.

Usage:  python tests/golden/make_golden_full4o.py        (~1-2 min of reference time on 8 threads)
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
    clip.load_state_dict(synth.outlier_edit(synth.synth_clip_state_dict(cfg, seed=111), cfg, seed=111), strict=True)
    model = A.AdaptedCLIP(clip, relu=False).eval()
    model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
    model.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
    anchors = torch.from_numpy(np.load(os.path.join(HERE, "full.npz"))["full.anchors_bottle"])

    B = 4
    img = synth.synth_images(B, 518, seed=4111)
    out = {"full4o.seed": np.int64(4111)}
    gelu_max = []
    hooks = [blk.mlp.gelu.register_forward_hook(lambda m, i, o: gelu_max.append(float(o.abs().max())))
             for blk in clip.visual.transformer.resblocks]
    with torch.no_grad():
        seg, det, iq = model(img)
    for h in hooks:
        h.remove()
    assert iq is None and len(gelu_max) == 24
    out["full4o.gelu_absmax"] = np.array(gelu_max, dtype=np.float32)
    for i, s_ in enumerate(seg):
        MG.put(out, f"full4o.seg{i}", s_)
    out["full4o.det"] = det.numpy()
    tfb = anchors.unsqueeze(0).repeat(B, 1, 1)
    total = 0
    with torch.no_grad():
        for i, s_ in enumerate(seg):
            sc = 100.0 * torch.matmul(s_, tfb)
            pp = sc.permute(0, 2, 1).view(B, 2, 37, 37)
            pre = (pp[:, 1] + 1 - pp[:, 0]) / 2
            out[f"full4o.map_pre_blur{i}"] = pre.numpy()
            total = total + pre
        out["full4o.map_pre_blur_sum"] = total.numpy()
        pooled, taps = clip.encode_image(img, [6, 12, 18, 24])
    out["full4o.pooled"] = pooled.numpy()
    out["full4o.stream_absmax"] = np.array([float(t.abs().max()) for t in taps], dtype=np.float32)
    for k, t in zip((6, 12, 18, 24), taps):
        MG.put(out, f"full4o.tap{k}", t)
    np.savez_compressed(os.path.join(HERE, "full4o.npz"), **out)
    print("full4o fixtures written; stream |max| at the taps:", out["full4o.stream_absmax"],
          "GELU |max| per block:", np.round(out["full4o.gelu_absmax"], 1))


if __name__ == "__main__":
    main()
