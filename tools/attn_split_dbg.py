import os, sys, torch
sys.path.insert(0, "aa-clip-iqm_amd"); sys.path.insert(0, ".")
from aaclip_hip import _lib, engine, synth
from aaclip_hip._lib import F16X2
dev = torch.device("cuda:0")
lib = _lib.load()
def ref_attn(qkv, B, L, H, causal):
    D = H * 64
    q, k, v = qkv.double().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=torch.float64).triu_(1)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, D)
for cfg in [(2, 1370, 2, 0), (1, 640, 3, 1)]:
    B, L, H, causal = cfg
    D = H * 64
    qkv = synth.randn("t.attn", (B * L, 3 * D), 1.0, 3)
    qkv[:, :D] *= 0.6
    qd = engine.split16_rows(qkv.to(dev))
    ctx = torch.full((B * L, 4 * D), 0xAA, dtype=torch.uint8, device=dev)
    _lib.check(lib.aaclip_attention(F16X2, qd.data_ptr(), ctx.data_ptr(), B, L, H, causal, torch.cuda.current_stream().cuda_stream), "attention")
    f = qkv.clone(); f[:, 2 * D:] = f[:, 2 * D:].half().float()
    ref = ref_attn(f, B, L, H, causal)
    got = engine.join_split8(ctx, D)
    err = (got.double().cpu() - ref).abs()
    bound = 4e-4 + 1e-3 * ref.abs()
    ratio = (err / bound)
    top = torch.topk(ratio.flatten(), 5)
    print(os.environ.get("AACLIP_LIB", "default"), cfg, "max ratio", float(ratio.max()), "rms err", float(err.pow(2).mean().sqrt()))
    for r, i in zip(top.values, top.indices):
        row, col = int(i) // D, int(i) % D
        print(f"   row {row} (token {row % L}) col {col}: err {float(err.flatten()[i]):.3e} ref {float(ref.flatten()[i]):+.3f} ratio {float(r):.2f}")
