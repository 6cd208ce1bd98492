#!/usr/bin/env python3
"""Experiment: does running the batch as two half batches on two HIP streams beat one stream?

One stream serialises the kernels of a block, so every GEMM pays its partial last round of tiles (out_proj / c_proj:
1372 tiles on 256 CUs = 5.36 rounds run as 6) and every CU reaches its residual read-modify-write epilogue at the same
time.  Two independent half batches on two streams let the dispatcher fill one kernel's tail with the other's workgroups.

    python tools/two_stream_probe.py [--precision fp16] [--batch 64] [--steps 10]
Prints images/s for: one stream (B), two streams (B/2 each) started together, two streams with the second delayed.
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--split", default="", help="comma-separated part sizes instead of equal parts, e.g. 59,5 (a big part whose "
                    "GEMMs fill whole rounds of 256 tiles and a small one to fill the gaps)")
    args = ap.parse_args()
    import torch
    from aaclip_hip import synth
    from model.clip import create_model
    dev = torch.device("cuda", 0)
    cfg = synth.ClipCfg()
    clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=args.precision, force_image_size=518)
    clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
    clip = clip.to(dev).eval()
    B, P = args.batch, args.parts
    images = torch.randn(B, 3, 518, 518, device=dev)
    if args.split:
        sizes = [int(v) for v in args.split.split(",")]
        assert sum(sizes) == B
        parts = list(images.split(sizes))
        P = len(parts)
    else:
        parts = list(images.chunk(P))
    streams = [torch.cuda.Stream(dev) for _ in range(P)]

    def one():
        return clip.encode_image(images, [6, 12, 18, 24])

    def multi(delay_ms=0.0):
        outs = []
        for i, (s, x) in enumerate(zip(streams, parts)):
            with torch.cuda.stream(s):
                if i and delay_ms:
                    torch.cuda._sleep(int(delay_ms * 1e-3 * 2.0e9 * i))
                outs.append(clip.encode_image(x, [6, 12, 18, 24]))
        return outs

    def timed(fn, label):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / args.steps
        print(f"{label:46s} {dt * 1e3:8.2f} ms/step  {B / dt:8.1f} images/s", flush=True)
        return dt

    with torch.no_grad():
        ref = one()
        torch.cuda.synchronize(dev)
        got = multi()
        torch.cuda.synchronize(dev)
        same = torch.equal(torch.cat([g[0] for g in got]), ref[0])
        print(f"pooled outputs of the {P}-stream run bit-identical to the one-stream run: {same}")
        timed(one, f"one stream, B = {B}")
        timed(multi, f"{P} streams, B = {[int(x.shape[0]) for x in parts]}")
        timed(lambda: multi(1.2), f"{P} streams, later ones delayed 1.2 ms")
        timed(one, f"one stream, B = {B} (again)")


if __name__ == "__main__":
    main()
