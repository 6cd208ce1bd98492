#!/usr/bin/env python3
"""Where the fp16 path's map error comes from: a CPU experiment on the oracle's arithmetic (test infrastructure; not
collected by pytest).  The full-size adapted visual forward is run in fp32 with ONE class of values rounded through fp16
at a time -- the roundings the MFMA path performs: weights, the LayerNorm outputs feeding QKV / c_fc, q/k/v, the softmax
probabilities, the attention context, the GELU output, the adapter input, the head inputs -- and the pre-blur 4-level
map sum is compared with the unrounded run.  usage: python tests/error_budget.py [--batch 1]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "aa-clip-iqm_amd"), REPO]
import numpy as np
import torch
import torch.nn.functional as F
from aaclip_hip import synth
from oracle import aaclip_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
torch.set_num_threads(a.threads)
cfg = synth.ClipCfg()
sd0 = synth.synth_clip_state_dict(cfg, 111)
ia0 = synth.synth_image_adapter_state_dict(cfg, seed=111)
img = synth.synth_images(a.batch, 518, seed=4111)
anchors = torch.from_numpy(np.load(os.path.join(REPO, "tests", "golden", "full.npz"))["full.anchors_bottle"])
heads = cfg.vision.heads
r16 = lambda t: t.half().float()


def forward(on):
    """on: set of rounding classes that are active"""
    R = lambda name, t: r16(t) if name in on else t
    sd = {k: (R("W", v) if v.is_floating_point() and v.dim() >= 2 else v) for k, v in sd0.items()}
    ia = {k: R("W", v) for k, v in ia0.items()}
    x = O.visual_stem(img, sd)
    segs = []
    for i in range(24):
        p = f"visual.transformer.resblocks.{i}."
        h = R("A1", O.layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]))
        B, L, D = h.shape
        hd = D // heads
        qkv = h @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
        q, k, v = qkv.split(D, dim=-1)
        q = R("QKV", q * hd ** -0.5).view(B, L, heads, hd).transpose(1, 2)
        k = R("QKV", k).view(B, L, heads, hd).transpose(1, 2)
        v = R("QKV", v).view(B, L, heads, hd).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        pm = torch.exp(s - s.amax(-1, keepdim=True))
        ctx = (R("P", pm) @ v) / pm.sum(-1, keepdim=True)
        ctx = R("CTX", ctx.transpose(1, 2).reshape(B, L, D))
        x = x + ctx @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
        h = R("A2", O.layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]))
        g = R("G", O.gelu_erf(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"]))
        x = x + g @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]
        if i < 6:
            w = ia[f"layer_adapters.{i}.fc.0.weight"]
            aa = O.leaky_relu(R("AD", x) @ w.t())
            aa = aa * x.norm(dim=-1, keepdim=True) / aa.norm(dim=-1, keepdim=True)
            x = 0.1 * aa + 0.9 * x
        if (i + 1) in (6, 12, 18, 24):
            t = R("HEAD", O.layer_norm(x[:, 1:, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]))
            segs.append(F.normalize(t @ ia[f"seg_proj.{len(segs)}.fc.weight"].t(), dim=-1))
    total = 0
    tfb = anchors.unsqueeze(0).repeat(a.batch, 1, 1)
    for s_ in segs:
        sc = 100.0 * torch.matmul(s_, tfb)
        total = total + (sc[..., 1] + 1 - sc[..., 0]) / 2
    return total


with torch.no_grad():
    ref = forward(set()).double()
    classes = ["W", "A1", "QKV", "P", "CTX", "A2", "G", "AD", "HEAD"]
    print(f"B={a.batch}; map sum |ref| max {ref.abs().max():.2f}")
    tot_sq = 0.0
    for c in classes + ["ALL"]:
        on = set(classes) if c == "ALL" else {c}
        e = (forward(on).double() - ref).abs()
        rms = float(e.pow(2).mean().sqrt())
        if c != "ALL":
            tot_sq += rms * rms
        print(f"{c:5s} rounded through fp16: max err {float(e.max()):.2e}  rms {rms:.2e}")
    print(f"root-sum-square of the single classes: rms {tot_sq ** 0.5:.2e}")
