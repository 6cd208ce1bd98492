"""Calibration only: what the vendor libraries reach on this path's GEMM / attention shapes.

Not part of the product path (which never calls hipBLASLt); the numbers give DESIGN.md a second
reference point beside the MFMA peak.  Run on the GPU box: python tools/bench_library.py
"""
import torch


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    dev = "cuda:0"
    M = 64 * 1370
    shapes = {"qkv": (1024, 3072), "out_proj": (1024, 1024), "c_fc": (1024, 4096), "c_proj": (4096, 1024)}
    for name, (K, N) in shapes.items():
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(N, K, device=dev, dtype=torch.float16)
        bias = torch.randn(N, device=dev, dtype=torch.float16)
        ms = timeit(lambda: torch.nn.functional.linear(a, w, bias))
        print(f"linear {name:9s} M={M} K={K} N={N}: {ms:.3f} ms -> {2 * M * K * N / ms / 1e9:.0f} TF", flush=True)
    q = torch.randn(64, 16, 1370, 64, device=dev, dtype=torch.float16)
    k = torch.randn_like(q)
    v = torch.randn_like(q)
    try:
        ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
        fl = 4 * 64 * 16 * 1370 * 1370 * 64
        print(f"sdpa B=64 H=16 L=1370 D=64: {ms:.3f} ms -> {fl / ms / 1e9:.0f} TF", flush=True)
    except Exception as e:  # noqa: BLE001
        print("sdpa failed:", e)


if __name__ == "__main__":
    main()
