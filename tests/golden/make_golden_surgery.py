#!/usr/bin/env python3
"""Golden vectors of the "CLIP surgery" tap path, produced by RUNNING THE REFERENCE
(reference model/transformer.py:102-152 Attention, :406-425 DAPM_replace; consumer train.py:75-85).

Same rules as make_golden.py, whose stubs and weight generator it reuses: runs only in the build
container, imports the reference, feeds it build-owned deterministic weights and records outputs
into tests/golden/surgery.npz.  Usage:  python tests/golden/make_golden_surgery.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

synth = G.synth


def main():
    torch.manual_seed(0)
    A, C, M, TK, FU, CONST = G._stub_and_import_reference()
    tcfg = synth.tiny_cfg()

    def build():
        m = M.CLIP(
            embed_dim=tcfg.embed_dim,
            vision_cfg=dict(image_size=tcfg.image_size, layers=tcfg.vision.layers, width=tcfg.vision.width,
                            patch_size=tcfg.patch_size, head_width=64),
            text_cfg=dict(context_length=tcfg.context_length, vocab_size=tcfg.vocab_size, width=tcfg.text.width,
                          heads=tcfg.text.heads, layers=tcfg.text.layers),
        ).eval()
        m.load_state_dict(synth.synth_clip_state_dict(tcfg, seed=7), strict=True)
        return m

    out = {}
    for B in (3, 1):
        img = synth.synth_images(B, tcfg.image_size, seed=7)
        for dpam in (2, 3):
            surgery = build()
            surgery.visual.DAPM_replace(DPAM_layer=dpam)
            with torch.no_grad():
                pooled, taps = surgery.encode_image(img, [1, 2, 3])
                out[f"b{B}.dpam{dpam}.pooled"] = pooled.numpy()
                for i, t in enumerate(taps):
                    out[f"b{B}.dpam{dpam}.tap{i + 1}"] = t.numpy()
                if B == 3 and dpam == 3:
                    # stage-1 feature chain of train.py:75-85
                    plain = build()
                    cls_token, _ = plain.encode_image(img, [])
                    cls_token = cls_token / cls_token.norm(dim=-1, keepdim=True)
                    feats = [surgery.visual.ln_post(t[:, 1:, :]) for t in taps]
                    feats = [t @ surgery.visual.proj for t in feats]
                    feats = [t / t.norm(dim=-1, keepdim=True) for t in feats]
                    feats = [t + cls_token.unsqueeze(1) for t in feats]
                    for i, t in enumerate(feats):
                        out[f"train_feat{i + 1}"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "surgery.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
