#!/usr/bin/env python3
"""Golden vectors of the IQM side branch produced by RUNNING THE REFERENCE (build container only).

The reference evaluates this branch with weights it never checkpoints (train.py:225-229 saves image_adapter only) and
creates two of its Linear layers lazily with fresh random weights inside forward (model/adapter.py:213-218,241-243).
To pin the arithmetic anyway, this script gives the reference's own modules DETERMINISTIC weights: the two lazy
layers are created before the call (the forward keeps them because their in_features match) and every parameter of
the branch is loaded from aaclip_hip.synth.synth_iqm_state_dict through the reference's own load_state_dict
(strict=False only because the CLIP / adapter keys are loaded separately, as in make_golden.py).

transformers 5.15 no longer ships PreTrainedModel.get_head_mask, which model/iqm.py:644 calls with head_mask=None:
the shim below restores its documented behaviour for that case ([None] * num_hidden_layers), as SURVEY 8(c) notes.

Recorded (full-size ViT-L/14@518, B = 4, the images and anchors of full4.npz):
  * iqm_outputs.last_hidden_state [4, 2, 768]                       (model/adapter.py:257-269)
  * the IQM anomaly maps of test_last.py:102-138: per level sigmoid(cos_abnormal - cos_normal) on the 37x37 grid,
    their bilinear (align_corners=False) 518x518 upsampling sampled, and the level sum
Usage:  python tests/golden/make_golden_iqm.py       (~40 s of reference time)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

synth = MG.synth


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    A, C, M, TK, FU, CONST = MG._stub_and_import_reference()
    from transformers.modeling_utils import PreTrainedModel
    if not hasattr(PreTrainedModel, "get_head_mask"):
        PreTrainedModel.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n
    cfg = synth.ClipCfg()
    clip = C.create_model("ViT-L-14-336", img_size=518, pretrained=None, force_image_size=518)
    clip.load_state_dict(synth.synth_clip_state_dict(cfg, seed=111), strict=True)
    model = A.AdaptedCLIP(clip, relu=False).eval()
    model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
    model.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
    model.visual_feature_proj = torch.nn.Linear(768, 768)
    model.text_feature_proj = torch.nn.Linear(2, 768)
    isd = synth.synth_iqm_state_dict(cfg, seed=111)
    missing, unexpected = model.load_state_dict(isd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("clipmodel.", "image_encoder.", "image_adapter.", "text_adapter.")) for k in missing), \
        [k for k in missing if not k.startswith(("clipmodel.", "image_encoder.", "image_adapter.", "text_adapter."))][:5]
    model.eval()

    g4 = np.load(os.path.join(HERE, "full4.npz"))
    anchors = torch.from_numpy(np.load(os.path.join(HERE, "full.npz"))["full.anchors_bottle"])
    B = 4
    img = synth.synth_images(B, 518, seed=int(g4["full4.seed"]))
    tfb = anchors.unsqueeze(0).repeat(B, 1, 1)          # [B, 768, 2] as test_last.py:84 builds it
    with torch.no_grad():
        seg, det, iq = model(img, text_embeddings=tfb)
    h = iq.last_hidden_state
    assert h.shape == (B, 2, 768)
    assert model.visual_feature_proj.in_features == 768 and model.text_feature_proj.in_features == 2
    out = {"iqm.last_hidden_state": h.numpy(), "iqm.seed": np.int64(111)}
    # seg tokens must be the ones full4.npz already holds (the branch does not change the main path)
    assert np.allclose(seg[3].reshape(-1)[torch.from_numpy(g4["full4.seg3.idx"])].numpy(), g4["full4.seg3.val"], atol=1e-6)
    norm_q, abn_q = h[:, 0, :], h[:, 1, :]
    total = 0
    for i, f in enumerate(seg):                          # test_last.py:108-138
        ns = F.cosine_similarity(f, norm_q.unsqueeze(1), dim=-1)
        an = F.cosine_similarity(f, abn_q.unsqueeze(1), dim=-1)
        pred = torch.sigmoid(an - ns).view(B, 1, 37, 37)
        out[f"iqm.grid{i}"] = pred[:, 0].numpy()
        up = F.interpolate(pred, size=(518, 518), mode="bilinear", align_corners=False)
        MG.put(out, f"iqm.map{i}", up[:, 0], stride=4999)
        total = total + up[:, 0]
    MG.put(out, "iqm.map_sum", total, stride=4999)
    out["iqm.map_sum_full0"] = total[0].numpy().astype(np.float32)[::7, ::7]     # a dense sub-grid of image 0
    np.savez_compressed(os.path.join(HERE, "iqm.npz"), **out)
    print("iqm fixtures written; |h| max", float(h.abs().max()), "grid range", float(out["iqm.grid0"].min()),
          float(out["iqm.grid0"].max()))


if __name__ == "__main__":
    main()
