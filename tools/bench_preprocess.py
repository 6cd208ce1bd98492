"""Throughput of the pre-processing kernel (uint8 HWC frames -> normalised fp32 NCHW at 518x518)
against its HBM roofline, next to Pillow on one host core.  python tools/bench_preprocess.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aa-clip-iqm_amd"))
from aaclip_hip import engine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    S = 518
    for (B, H, W) in [(64, 1024, 1024), (64, 700, 700), (64, 518, 518), (64, 256, 256)]:
        src = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)
        for _ in range(3):
            engine.preprocess(src, S)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            engine.preprocess(src, S)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        nbytes = B * H * W * 3 + B * 3 * S * S * 4
        print(f"B={B} {H}x{W} -> {S}: {ms:.3f} ms  {B / ms * 1e3:.0f} img/s  {nbytes / ms / 1e6:.0f} GB/s algorithmic "
              f"({nbytes / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
    from PIL import Image
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8))
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        arr = np.asarray(img.resize((S, S), Image.BICUBIC))
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().float().div(255)
        t.sub_(torch.tensor(engine.CLIP_MEAN).view(-1, 1, 1)).div_(torch.tensor(engine.CLIP_STD).view(-1, 1, 1))
    dt = (time.perf_counter() - t0) / n
    print(f"Pillow + torch on one host core, 1024x1024 -> {S}: {dt * 1e3:.2f} ms/image = {1 / dt:.0f} img/s")


if __name__ == "__main__":
    main()
