#!/usr/bin/env python3
"""Times the split-fp16 (AACLIP_F16X2) GEMM kernel on the tower's four shapes: ms per launch and algorithmic TFLOP/s.
Environment: VARIANTS=80,81 (kernel A/B with a bit-identity check), ONLY=c_fc,qkv (subset of the shapes), MROWS=<rows>.
AACLIP_LIB selects an experiment build of the library.  python tools/bench_split_gemm.py [--exact]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import _lib, engine

def main():
    exact = "--exact" in sys.argv      # (kept for old command lines; the weight format follows the library call)
    variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = int(os.environ.get("MROWS", 64 * 1370))     # MROWS=81920: 320 row tiles, every shape an exact number of rounds
    st = torch.cuda.current_stream(dev).cuda_stream
    only = os.environ.get("ONLY", "")
    for name, (K, N, epi) in {"qkv": (1024, 3072, 0), "out_proj": (1024, 1024, 2), "c_fc": (1024, 4096, 1),
                              "c_proj": (4096, 1024, 2)}.items():
        if only and name not in only.split(","):
            continue
        A = torch.randint(0, 255, (M, 4 * K), dtype=torch.uint8, device=dev)
        A[:, 1::2][:, :K] &= 0x3B          # keep the fp16 plane's exponents moderate (finite values)
        W = torch.randint(0, 255, (N, 4 * K), dtype=torch.uint8, device=dev)
        W[:, 1::2][:, :K] &= 0x3B
        bias = torch.zeros(N, device=dev)
        out = torch.zeros(M, N, dtype=torch.float32, device=dev) if epi == 2 else torch.empty(M, 4 * N, dtype=torch.uint8, device=dev)
        ldc = N if epi == 2 else 2 * N
        def run():
            _lib.check(lib.aaclip_gemm(_lib.F16X2, epi, A.data_ptr(), 2 * K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), ldc,
                                       M, N, K, 0, 0, 1.0, st), "gemm")
        first = None
        for v in variants:
            assert lib.aaclip_set_gemm_variant(v) == 0, v
            if epi == 2:
                out.zero_()
            run()
            torch.cuda.synchronize()
            got = out.clone()
            if first is None:
                first = got
            else:   # every 256-family kernel computes the same sums in the same order: bit-identical outputs
                same = torch.equal(got, first)
                print(f"{name}: variant {v} vs {variants[0]}: {'bit-identical' if same else 'DIFFERENT'}", flush=True)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"{name:9s} K={K} N={N} variant {v}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s algorithmic", flush=True)
        lib.aaclip_set_gemm_variant(0)

if __name__ == "__main__":
    main()
