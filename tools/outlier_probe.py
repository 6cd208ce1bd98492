import os, sys
sys.path.insert(0, "aa-clip-iqm_amd"); sys.path.insert(0, "tools")
import torch
from aaclip_hip import engine, _lib
from aaclip_hip._lib import F16
from model.transformer import ResidualAttentionBlock
import importlib.util
spec = importlib.util.spec_from_file_location("sb", "tools/stress_blocks.py")
# reuse ref_block by exec'ing the function definitions only
src = open("tools/stress_blocks.py").read()
ns = {"torch": torch}
exec(src[src.index("def ln("):src.index("n = 0\nfor _ in range(10):")], ns)
ref_block = ns["ref_block"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
D, H, L, B = 1024, 16, 1370, 4
blk = ResidualAttentionBlock(D, H).to(dev)
with torch.no_grad():
    for p in blk.parameters():
        if p.dim() > 1: p.normal_(0, 0.7 * p.shape[1] ** -0.5)
        else: p.normal_(0, 0.3)
    blk.ln_1.weight.add_(1.0); blk.ln_2.weight.add_(1.0)
lib = _lib.load()
for name, outl in (("no outliers", []), ("outliers +80/-60", [(7, 80.0), (300, -60.0)]), ("outliers +300", [(7, 300.0)]),
                   ("row mean 5", "mean")):
    x0 = torch.randn(B * L, D, device=dev)
    if outl == "mean":
        x0 += 5.0
    else:
        for c, v in outl:
            x0[:, c] += v * (1 + 0.1 * torch.randn(B * L, device=dev))
    ref = ref_block(x0.double(), blk, B, L, H, False, False, None, 0.1)
    res = {}
    for flag, tag in ((0, "fold"), (1 << 17, "ln pass")):
        lib.aaclip_set_gemm_variant(flag)
        x = x0.clone()
        with torch.no_grad():
            engine.run_blocks(x, [blk, ], B, L, H, F16)
        d = (x.double() - ref)
        # error of the block's UPDATE (output minus input), where ln_2/c_fc matter
        res[tag] = (d.abs().max().item(), d.abs().mean().item())
    lib.aaclip_set_gemm_variant(0)
    print(f"{name:18s} fold: max {res['fold'][0]:.2e} mean {res['fold'][1]:.2e} | ln pass: max {res['ln pass'][0]:.2e} mean {res['ln pass'][1]:.2e}")
