#!/usr/bin/env python3
"""Full-size B = 4 record from the REFERENCE with fp16-NATIVE CLIP weights (build container only).

OpenAI's ViT-L/14-336 checkpoint stores its weights in fp16; the reference converts them to fp32 on load
(model/clip.py:88), so every CLIP weight it computes with is exactly representable in fp16 and the fp16 MFMA path's
weight conversion is lossless.  The seeded weights of the other fixtures are fp32 random numbers: rounding THEM to
fp16 is an error source the real deployment does not have.  This record repeats make_golden_full4.py with the CLIP
state dict rounded through fp16 first (adapters stay fp32: they are trained in fp32), so that the fp16 path's map error
can be measured for the deployment case (tests/test_gpu_configs.py::test_full_b4_fp16_native_weights).

Usage:  python tests/golden/make_golden_full4h.py        (~1 min of reference time on 8 threads)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

synth = MG.synth


def fp16_native(sd):
    return {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    A, C, M, TK, FU, CONST = MG._stub_and_import_reference()
    cfg = synth.ClipCfg()
    clip = C.create_model("ViT-L-14-336", img_size=518, pretrained=None, force_image_size=518)
    clip.load_state_dict(fp16_native(synth.synth_clip_state_dict(cfg, seed=111)), strict=True)
    model = A.AdaptedCLIP(clip, relu=False).eval()
    model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
    anchors = torch.from_numpy(np.load(os.path.join(HERE, "full.npz"))["full.anchors_bottle"])
    B = 4
    img = synth.synth_images(B, 518, seed=4111)
    out = {"full4h.seed": np.int64(4111)}
    with torch.no_grad():
        seg, det, _ = model(img)
        for i, s_ in enumerate(seg):
            MG.put(out, f"full4h.seg{i}", s_)
        out["full4h.det"] = det.numpy()
        tfb = anchors.unsqueeze(0).repeat(B, 1, 1)
        total = 0
        for i, s_ in enumerate(seg):
            sc = 100.0 * torch.matmul(s_, tfb)
            pp = sc.permute(0, 2, 1).view(B, 2, 37, 37)
            pre = (pp[:, 1] + 1 - pp[:, 0]) / 2
            out[f"full4h.map_pre_blur{i}"] = pre.numpy()
            total = total + pre
        out["full4h.map_pre_blur_sum"] = total.numpy()
    np.savez_compressed(os.path.join(HERE, "full4h.npz"), **out)
    print("full4h fixtures written:", sorted(k for k in out if k.endswith(".shape")))


if __name__ == "__main__":
    main()
