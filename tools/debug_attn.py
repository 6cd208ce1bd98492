import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
import torch
from aaclip_hip import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def ref(qkv, B, L, H):
    D = 64 * H
    q, k, v = qkv.float().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    return (a @ v).permute(0, 2, 1, 3).reshape(B * L, D)
for (B, L, H) in [(1, 512, 1), (1, 1370, 1), (2, 1370, 2), (1, 640, 1)]:
    torch.manual_seed(0)
    D = 64 * H
    qkv = torch.randn(B * L, 3 * D, device=dev)
    qkv[:, :D] *= 0.5
    qkv = qkv.half()
    ctx = torch.full((B * L, D), float("nan"), device=dev, dtype=torch.float16)
    _lib.check(lib.aaclip_attention(_lib.F16, qkv.data_ptr(), ctx.data_ptr(), B, L, H, 0, st))
    torch.cuda.synchronize()
    r = ref(qkv, B, L, H)
    c = ctx.float()
    nan = torch.isnan(c)
    err = (c - r).abs()
    err[nan] = 0
    print(f"B={B} L={L} H={H}: nan frac {nan.float().mean().item():.4f}; max err (finite) {err.max().item():.3e}")
    if nan.any():
        rows = nan.any(dim=1).nonzero().flatten()
        print("  nan rows: first", rows[:8].tolist(), "last", rows[-8:].tolist(), "count", rows.numel())
        cols = nan.any(dim=0).nonzero().flatten()
        print("  nan cols:", cols[:8].tolist(), "...", cols.numel())
    big = (err > 1e-2).any(dim=1).nonzero().flatten()
    if big.numel():
        print("  rows with err>1e-2: count", big.numel(), "first", big[:10].tolist(), "last", big[-5:].tolist())
        rr = big[0].item()
        print("  row", rr, "got", c[rr, :6].tolist(), "ref", r[rr, :6].tolist())
