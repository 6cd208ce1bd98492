"""GPU parity tests of the split-fp16 arithmetic (AACLIP_F16X2, precision='fp16x2'; include/aaclip.h, csrc/common.h):
every matrix-product operand is fp16(v) plus a correction for v - fp16(v).  GEMMs accumulate Ah.Wh on the fp16 MFMAs and
the two correction products Al8.Wh8 + Ah8.Wl8 on the block-scaled e4m3 MFMAs (split8 rows); attention runs q.k^T as
three and p.v as two fp16 products on fp16 hi + lo pairs (split16 rows).

The mode exists to put the 16-bit MFMA path inside BASELINE.json's tolerance (1e-3 abs + 1e-2 rel vs the fp32
reference) on taps and anomaly maps.  Kernel-level bounds here: ~1e-4 relative on a GEMM result (plain fp16: 1.5e-3;
the correction terms carry 4 significant bits, i.e. ~2^-15 per operand), 4e-4 on an attention context.
Full-model checks against the reference's golden vectors are in tests/test_gpu_configs.py (the `fp16x2` cases).
"""
import pytest
import torch

from aaclip_hip import _lib, engine, synth
from aaclip_hip._lib import F16, F16X2, F32
from oracle import aaclip_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def assert_close(a, b, atol, rtol, what=""):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite output"
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} outside {atol}+{rtol}*|ref|; "
                           f"max err {err.max().item():.3e} at ref {b.flatten()[err.argmax()].item():.3e}")


def join16(t, C):
    """split16 rows [R, 2C] fp16 -> fp64 values hi + lo"""
    return t[:, :C].double().cpu() + t[:, C:2 * C].double().cpu()


join8 = engine.join_split8      # split8 rows -> fp64 hi + lo8 * 2^-10


def test_split_rows_formats(dev):
    """engine.split_rows / split16_rows: what a product sees of a value -- hi + lo -- is within 2^-21 (split16) and
    2^-15 (split8: the correction has 4 significant bits) of the fp32 value."""
    x = synth.randn("t.split.x", (64, 256), 3.0, 1).to(dev)
    s16 = engine.split16_rows(x)
    assert s16.shape == (64, 512) and s16.dtype == torch.float16
    err = (join16(s16, 256) - x.double().cpu()).abs()
    assert float((err / (x.double().cpu().abs() * 2.0 ** -21 + 2.0 ** -24)).max()) <= 1.0
    s8 = engine.split_rows(x)
    assert s8.shape == (64, 1024) and s8.dtype == torch.uint8
    err = (join8(s8, 256) - x.double().cpu()).abs()
    assert float((err / (x.double().cpu().abs() * 2.0 ** -15 + 2.0 ** -19)).max()) <= 1.0
    w = engine.split_rows(x, weight=True)
    wx = engine.split_rows(x.half().float(), weight=True, exact=True)
    assert w.shape == (64, 1024) and wx.shape == (64, 768)
    assert torch.equal(w[:, :512].cpu(), s8[:, :512].cpu())            # the fp16 plane is the same


@pytest.mark.parametrize("D", [256, 768, 1024])
def test_layernorm_split_output(dev, D):
    lib = _lib.load()
    x = synth.randn("t.ln.x", (37, D), 3.0, 1, mean=0.7)
    w, b = synth.randn("t.ln.w", (D,), 0.2, 1, 1.0), synth.randn("t.ln.b", (D,), 0.2, 1)
    out = torch.full((37, 4 * D), 0xAA, dtype=torch.uint8, device=dev)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    _lib.check(lib.aaclip_layernorm(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), F16X2, 37, D, 1e-5,
                                    stream(dev)))
    ref = O.layer_norm(x.double(), w.double(), b.double())
    assert_close(join8(out, D), ref, 6e-6, 2.0 ** -14, f"layernorm split8 {D}")
    # the kernel's planes are exactly what the host-side split of the SAME fp32 values gives (hi, lo8, hi8 bytes)
    hi = out[:, : 2 * D].cpu().contiguous().view(torch.float16)
    want = engine.split_rows(hi.float() + 0.0)            # hi plane of a value that is exact in fp16
    assert torch.equal(want[:, : 2 * D].cpu(), out[:, : 2 * D].cpu())
    hi8 = out[:, 3 * D:].cpu().contiguous().view(torch.float8_e4m3fn).double()
    assert float(((hi8 - ref).abs() / (ref.abs() * 2.0 ** -4 + 2.0 ** -9)).max()) <= 1.0


def _gemm(lib, dev, epi, A, W, bias, out, K, act=0, scale_cols=0, scale=1.0):
    """A: split8 rows (uint8 [M, 4K]); W: split8 weight rows; out: fp32 [M, N], or uint8 [M, 4N] split rows"""
    M, N = A.shape[0], W.shape[0]
    ldc = out.shape[1] if out.dtype == torch.float32 else 2 * N
    _lib.check(lib.aaclip_gemm(F16X2, epi, A.data_ptr(), 2 * K, W.data_ptr(), None if bias is None else bias.data_ptr(),
                               out.data_ptr(), ldc, M, N, K, act, scale_cols, scale, stream(dev)), "gemm")


def _planes(hi=None, p8a=None, p8b=None, rows=0, K=0):
    """hand-built split8 rows from explicit planes: fp16 values, bytes of plane 1, bytes of plane 2 (None = zeros)"""
    z16 = torch.zeros(rows, K, dtype=torch.float16)
    z8 = torch.zeros(rows, K, dtype=torch.uint8)
    parts = [(hi if hi is not None else z16).contiguous().view(torch.uint8).reshape(rows, 2 * K),
             p8a if p8a is not None else z8, p8b if p8b is not None else z8]
    return torch.cat(parts, dim=1).contiguous()


def _e4m3(t):
    return t.float().to(torch.float8_e4m3fn).view(torch.uint8)


@pytest.mark.parametrize("M", [200, 4300])   # 128-tile kernel / 256-tile kernel
def test_gemm_split_exact_integers(dev, M):
    """Small integers are exact in fp16 and in e4m3: any wrong lane / fragment / virtual-tile / scale mapping shows up
    bit-for-bit.  Four runs: ordinary split8 operands (corrections are zero); the integers carried ONLY by the
    activation's lo8 plane against the weight's e4m3 hi plane (tile T1, scale 2^-(10+6)); ONLY by the activation's hi8
    plane against the weight's lo8 plane (tile T2, scale 2^-17); and a weight passed in its 3-plane 'exact' form."""
    lib = _lib.load()
    N, K = 256, 256
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float()
    A[:, 0] += torch.arange(M).float() % 5
    W[:, 1] += torch.arange(N).float() % 3
    ref = A.double() @ W.double().t()
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    As, Ws = engine.split_rows(A.to(dev)), engine.split_rows(W.to(dev), weight=True)
    _gemm(lib, dev, _lib.EPI_ACT_F32, As, Ws, None, out, K)
    assert torch.equal(out.cpu().double(), ref)
    # T1 only: value = lo8 * 2^-10 on the activation side, e4m3(W * 2^6) on the weight side
    A1 = _planes(p8a=_e4m3(A), rows=M, K=K).to(dev)
    W1 = _planes(p8a=_e4m3(W * 64), rows=N, K=K).to(dev)
    _gemm(lib, dev, _lib.EPI_ACT_F32, A1, W1, None, out, K)
    assert torch.equal(out.cpu().double() * 1024.0, ref)
    # T2 only: e4m3(A) on the activation side, weight lo plane = W * 2^-17 stored as e4m3(W)
    A2 = _planes(p8b=_e4m3(A), rows=M, K=K).to(dev)
    W2 = _planes(p8b=_e4m3(W), rows=N, K=K).to(dev)
    _gemm(lib, dev, _lib.EPI_ACT_F32, A2, W2, None, out, K)
    assert torch.equal(out.cpu().double() * 131072.0, ref)


@pytest.mark.parametrize("shape", [(1370, 1024, 1024), (300, 128, 128), (77, 768, 3072), (129, 384, 640),
                                   (5000, 768, 1024),    # large M, ragged last tile: the 256-tile kernel
                                   (4500, 1024, 4096),   # c_proj's shape: 128 virtual tiles
                                   (4200, 256, 640)])    # patch embed's K: 5 K-tile pairs
def test_gemm_split_epilogues(dev, shape):
    lib = _lib.load()
    M, N, K = shape
    A = synth.randn("t.g.a", (M, K), 1.0, 2)
    W = synth.randn("t.g.w", (N, K), K ** -0.5, 2)
    bias = synth.randn("t.g.b", (N,), 0.5, 2)
    Ad, Wd, bd = engine.split_rows(A.to(dev)), engine.split_rows(W.to(dev), weight=True), bias.to(dev)
    acc = A.double() @ W.double().t()
    # each correction term carries 4 significant bits: ~2^-15 relative per operand, random over K -> measured ~1e-5 rms,
    # 1.5e-4 max on rows of |a|.|w| ~ 1 (plain fp16: 6e-4 rms); asserted at 3e-4 + 1e-4 |ref|
    ea, er = 3e-4, 1e-4
    out = torch.full((M, 4 * N), 0xAA, dtype=torch.uint8, device=dev)
    _gemm(lib, dev, _lib.EPI_BIAS, Ad, Wd, bd, out, K, scale_cols=64, scale=0.125)       # -> split16 rows
    ref = acc + bias.double()
    ref[:, :64] *= 0.125
    assert_close(join16(out.view(torch.float16), N), ref, ea, er, "bias")
    _gemm(lib, dev, _lib.EPI_BIAS_GELU, Ad, Wd, bd, out, K)                                  # -> split8 rows
    assert_close(join8(out, N), O.gelu_erf(acc + bias.double()), ea, er + 2.0 ** -14, "gelu")
    x0 = synth.randn("t.g.x", (M, N), 2.0, 2)
    xd = x0.to(dev)
    _gemm(lib, dev, _lib.EPI_BIAS_RESID, Ad, Wd, bd, xd, K)
    assert_close(xd, x0.double() + acc + bias.double(), ea, er, "resid")
    o32 = torch.empty(M, N, dtype=torch.float32, device=dev)
    _gemm(lib, dev, _lib.EPI_ACT_F32, Ad, Wd, None, o32, K, act=1)
    assert_close(o32, O.leaky_relu(acc), ea, er, "leaky")
    # a weight that is exact in fp16 (its lo plane is all zero): same function of the rounded weight.  (The 3-plane
    # 'exact' form, which skips the weight-lo tile, is selected by aaclip_block_weights.exact16 and is exercised by
    # test_full_b4_fp16_native_weights; the generic aaclip_gemm entry point always takes 4-plane weights.)
    Wh = W.half().float()
    W4 = engine.split_rows(Wh.to(dev), weight=True)
    _gemm(lib, dev, _lib.EPI_ACT_F32, Ad, W4, None, o32, K)
    assert_close(o32, A.double() @ Wh.double().t(), ea, er, "fp16-exact weight, 4 planes")


def test_gemm_split_ragged_shapes_sweep(dev):
    """Seeded sweep over row counts around the kernel switch (M = 4096) and ragged last tiles (the 256-tile split kernel
    reads rows past M through the buffer descriptor's bounds instead of clamping), K tile-pair counts 1...9, N of one to
    five 128- / 256-column tiles: fp32-out result against fp64, and rows [M, M + 3) of an over-allocated output untouched."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(20260104)
    cases = [(4095, 256, 128), (4096, 256, 128), (4097, 512, 256), (4351, 256, 1152), (4352, 768, 384), (4609, 1280, 640),
             (8191, 256, 512), (1, 128, 128), (127, 384, 256), (129, 128, 896), (300, 640, 1024)]
    for _ in range(6):
        M = int(torch.randint(4096, 9000, (1,), generator=g))
        cases.append((M, 256 * int(torch.randint(1, 4, (1,), generator=g)), 128 * int(torch.randint(1, 10, (1,), generator=g))))
    for M, N, K in cases:
        A = synth.randn(f"t.sw.a{M}", (M, K), 1.0, 3)
        W = synth.randn(f"t.sw.w{N}", (N, K), K ** -0.5, 3)
        Ad, Wd = engine.split_rows(A.to(dev)), engine.split_rows(W.to(dev), weight=True)
        out = torch.full((M + 3, N), 7.5, dtype=torch.float32, device=dev)
        _gemm(lib, dev, _lib.EPI_ACT_F32, Ad, Wd, None, out[:M], K)
        assert_close(out[:M], A.double() @ W.double().t(), 3e-4, 1e-4, f"sweep {(M, N, K)}")
        assert bool((out[M:] == 7.5).all()), f"rows past M written for {(M, N, K)}"


def _attn_ref(qkv, B, L, H, causal):
    D = H * 64
    q, k, v = qkv.double().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + O.causal_mask(L, torch.float64)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, D)


@pytest.mark.parametrize("log2q", [0, 1])
@pytest.mark.parametrize("cfg", [(2, 1370, 2, 0), (3, 77, 4, 1), (1, 50, 1, 0), (2, 130, 2, 1), (1, 64, 1, 0),
                                 (1, 129, 1, 1), (5, 1, 2, 0), (26, 3, 4, 0), (7, 2, 1, 1), (1, 640, 3, 1)])
def test_attention_split(dev, cfg, log2q):
    """q.k^T on 3 products (q and k with their lo halves), p.v on one (p and v in fp16: v's lo half is read by nobody,
    see attention.hip).  Against fp64 attention of the same q, k and the fp16-rounded v the context is within ~3e-4
    -- the fp16 rounding of p averaged over the row -- where plain fp16 is asserted at 3e-3; the rounding of v itself
    is a property of the format, bounded separately below.  Rows shorter than 512 keys use v's lo half as well."""
    lib = _lib.load()
    B, L, H, causal = cfg
    D = H * 64
    qkv = synth.randn("t.attn", (B * L, 3 * D), 1.0, 3)
    qkv[:, :D] *= 0.6 * (1.4426950408889634 if log2q else 1.0)
    qd = engine.split16_rows(qkv.to(dev))
    ctx = torch.full((B * L, 4 * D), 0xAA, dtype=torch.uint8, device=dev)      # split8 rows out
    fn = lib.aaclip_attention_log2q if log2q else lib.aaclip_attention
    _lib.check(fn(F16X2, qd.data_ptr(), ctx.data_ptr(), B, L, H, causal, stream(dev)), "attention")
    f = qkv.clone()
    if log2q:
        f[:, :D] *= 0.6931471805599453
    exact = _attn_ref(f, B, L, H, causal)
    if L >= 512:      # long rows: v in fp16 (short ones keep v's lo half)
        f[:, 2 * D:] = f[:, 2 * D:].half().float()
    ref = _attn_ref(f, B, L, H, causal)
    assert_close(join8(ctx, D), ref, 4e-4, 1e-3, f"attention split {cfg}")
    # v in fp16: a row of the context moves by at most 2^-11 of the largest |v| it averages
    assert_close(join8(ctx, D), exact, 4e-4 + 2.0 ** -11 * float(qkv[:, 2 * D:].abs().max()), 1e-3, f"vs unrounded v {cfg}")


@pytest.mark.parametrize("code", [F16, F16X2])
@pytest.mark.parametrize("L", [200, 700])      # 128-query kernel / long-sequence kernel (plain fp16)
@pytest.mark.parametrize("causal", [0, 1])
def test_attention_first_tile_far_below_zero(dev, code, L, causal):
    """Every score of a row's FIRST key tile at about -150 in log2 units: the re-base factor 2^150 of that tile
    overflows fp32, and multiplying the still-zero row sum and output by it gave NaN rows (0 * inf).  torch's softmax
    in the reference handles any finite logits (model/transformer.py:200)."""
    lib = _lib.load()
    H, D = 1, 64
    qkv = synth.randn("t.attn.neg", (L, 192), 0.3, 9)
    qkv[:, :64] *= 0.2
    qkv[:, 0] = 1.0                       # q[:, 0] = 1
    qkv[:, 64] = -150.0                   # k[:, 0]: every logit ~ -150 (log2 units), in every tile
    qkv[L - 3, 64] = 4.0                  # ... except one late key
    if code == F16:
        q = qkv.to(torch.float16)
        qd = q.to(dev)
        ctx = torch.full((L, D), float("nan"), dtype=torch.float16, device=dev)
        ref_in = q.float()
    else:
        qd = engine.split16_rows(qkv.to(dev))
        ctx = torch.full((L, 4 * D), 0xAA, dtype=torch.uint8, device=dev)
        ref_in = qkv.clone()
    _lib.check(lib.aaclip_attention_log2q(code, qd.data_ptr(), ctx.data_ptr(), 1, L, H, causal, stream(dev)))
    ref_in[:, :64] *= 0.6931471805599453
    if L >= 512:
        ref_in[:, 128:] = ref_in[:, 128:].half().float()
    ref = _attn_ref(ref_in, 1, L, H, causal)
    got = ctx.float() if code == F16 else join8(ctx, D)
    assert_close(got, ref, 3e-3 if code == F16 else 4e-4, 1e-2, f"first tile -150, L={L}, causal={causal}")


def _ref_block64(x, blk, B, L, H):
    """One residual attention block in fp64 (reference model/transformer.py:239-258), full attention."""
    D = x.shape[1]
    d = lambda t: t.detach().double()
    ln = lambda v, w, b: torch.nn.functional.layer_norm(v, (D,), d(w), d(b), 1e-5)
    h = ln(x, blk.ln_1.weight, blk.ln_1.bias)
    qkv = (h @ d(blk.attn.in_proj_weight).t() + d(blk.attn.in_proj_bias)).view(B, L, 3, H, 64)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    ctx = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B * L, D)
    x = x + ctx @ d(blk.attn.out_proj.weight).t() + d(blk.attn.out_proj.bias)
    h = ln(x, blk.ln_2.weight, blk.ln_2.bias)
    h = h @ d(blk.mlp.c_fc.weight).t() + d(blk.mlp.c_fc.bias)
    h = 0.5 * h * (1 + torch.erf(h / 2 ** 0.5))
    return x + h @ d(blk.mlp.c_proj.weight).t() + d(blk.mlp.c_proj.bias)


@pytest.mark.parametrize("shape", [(256, 520, 8), (512, 1370, 3)])
def test_block_long_rows_take_the_e4m3_attention_form(dev, shape):
    """Long rows on the 256-tile kernels: the QKV epilogue writes q and k as fp16 + [lo8 | hi8] e4m3 records and the
    attention kernel runs its two correction products on the 32x32x64 scaled MFMA (csrc/attention.hip, QK8).  Against
    fp64 on the same weights, and against the split16 form of the same block (forced by routing the products to the
    128-tile kernels): both inside the same bound, and their difference far below plain fp16's error."""
    from model.transformer import ResidualAttentionBlock
    D, L, B = shape
    H = D // 64
    torch.manual_seed(5)
    blk = ResidualAttentionBlock(D, H).to(dev)
    with torch.no_grad():
        for prm in blk.parameters():
            if prm.dim() > 1:
                prm.normal_(0, 0.7 * prm.shape[1] ** -0.5)
            else:
                prm.normal_(0, 0.3)
        blk.ln_1.weight.add_(1.0)
        blk.ln_2.weight.add_(1.0)
    x0 = torch.randn(B * L, D, device=dev)
    x0[:, 3] += 2.0
    ref = _ref_block64(x0.double(), blk, B, L, H)
    outs = {}
    lib = _lib.load()
    with torch.no_grad():
        for name, variant in (("qk8", 0), ("split16", 1)):
            assert lib.aaclip_set_gemm_variant(variant) == 0
            try:
                xa = x0.clone()
                engine.run_blocks(xa, [blk], B, L, H, F16X2, causal=False)
                outs[name] = xa
            finally:
                lib.aaclip_set_gemm_variant(0)
        x16 = x0.clone()
        engine.run_blocks(x16, [blk], B, L, H, F16, causal=False)
    e8 = float((outs["qk8"].double() - ref).abs().max())
    e16 = float((outs["split16"].double() - ref).abs().max())
    ef = float((x16.double() - ref).abs().max())
    print(f"one block D{D} L{L}: |err| vs fp64: e4m3 form {e8:.2e}, split16 form {e16:.2e}, plain fp16 {ef:.2e}")
    assert not torch.equal(outs["qk8"], outs["split16"])      # the two forms really are different code paths
    assert_close(outs["qk8"], ref, 4e-4, 5e-4, "e4m3 attention form vs fp64")
    assert_close(outs["split16"], ref, 4e-4, 5e-4, "split16 attention form vs fp64")
    assert e8 < 0.25 * ef and e16 < 0.25 * ef


# ------------------------------------------------------------------------------------------------------------------
# the 256-row-tile GEMM kernels (8 waves, 256 x 256, one tile per workgroup = 80 / 4 waves, 256 x 128, two workgroups per
# CU = 81 / the 8-wave kernel walking its tiles, one workgroup per CU, the next tile's first K tile fetched under the
# epilogue = 82, the default for split operands) compute the same sums in the same order: every output must be
# BIT-identical between aaclip_set_gemm_variant(80), (81) and (82).  Shapes with more than 256 tiles make 82 really walk.
# ------------------------------------------------------------------------------------------------------------------
def _variant(lib, v):
    assert lib.aaclip_set_gemm_variant(v) == 0, v


@pytest.mark.parametrize("shape", [(4300, 256, 128), (5480, 1024, 1024), (4097, 768, 384), (6000, 512, 2048),
                                   (40000, 1024, 256), (70001, 256, 384), (33000, 3072, 128)])
def test_gemm_half_tile_kernel_is_bit_identical_split(dev, shape):
    lib = _lib.load()
    M, N, K = shape
    A = synth.randn("t.h.a", (M, K), 1.0, 4)
    W = synth.randn("t.h.w", (N, K), K ** -0.5, 4)
    bias = synth.randn("t.h.b", (N,), 0.5, 4).to(dev)
    Ad, Wd = engine.split_rows(A.to(dev)), engine.split_rows(W.to(dev), weight=True)
    x0 = synth.randn("t.h.x", (M, N), 2.0, 4)
    res = {}
    try:
        for v in (80, 81, 82):
            _variant(lib, v)
            outs = []
            o16 = torch.full((M + 2, 4 * N), 0xAA, dtype=torch.uint8, device=dev)
            _gemm(lib, dev, _lib.EPI_BIAS, Ad, Wd, bias, o16[:M], K, scale_cols=64, scale=0.125)
            outs.append(o16.clone())
            _gemm(lib, dev, _lib.EPI_BIAS_GELU, Ad, Wd, bias, o16[:M], K)
            outs.append(o16.clone())
            xd = x0.to(dev)
            _gemm(lib, dev, _lib.EPI_BIAS_RESID, Ad, Wd, bias, xd, K)
            outs.append(xd)
            o32 = torch.full((M + 2, N), 7.5, dtype=torch.float32, device=dev)
            _gemm(lib, dev, _lib.EPI_ACT_F32, Ad, Wd, None, o32[:M], K, act=1)
            outs.append(o32)
            res[v] = outs
    finally:
        _variant(lib, 0)
    for a, b, c, what in zip(res[80], res[81], res[82], ("bias", "gelu", "resid", "leaky")):
        assert torch.equal(a, b), f"{what} {shape}: the half-tile kernel differs from the 8-wave kernel"
        assert torch.equal(a, c), f"{what} {shape}: the walking kernel differs from the 8-wave kernel"
    assert bool((res[82][3][M:] == 7.5).all()) and bool((res[82][0][M:] == 0xAA).all())     # rows past M untouched
    assert bool((res[81][3][M:] == 7.5).all()) and bool((res[81][0][M:] == 0xAA).all())     # rows past M untouched
    assert_close(res[81][3][:M], O.leaky_relu(A.double() @ W.double().t()), 3e-4, 1e-4, f"half-tile leaky {shape}")


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_gemm_half_tile_kernel_is_bit_identical_plain(dev, dtype):
    lib = _lib.load()
    code = {"fp16": F16, "bf16": _lib.BF16}[dtype]
    tdt = {"fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    for M, N, K in [(4300, 256, 128), (5480, 1024, 1024), (4500, 768, 3072), (4097, 3072, 768)]:
        A = synth.randn("t.hp.a", (M, K), 1.0, 4).to(dev).to(tdt)
        W = synth.randn("t.hp.w", (N, K), K ** -0.5, 4).to(dev).to(tdt)
        bias = synth.randn("t.hp.b", (N,), 0.5, 4).to(dev)
        x0 = synth.randn("t.hp.x", (M, N), 2.0, 4)
        res = {}
        try:
            for v in (80, 81):
                _variant(lib, v)
                outs = []
                for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU):
                    o = torch.zeros(M, N, dtype=tdt, device=dev)
                    _lib.check(lib.aaclip_gemm(code, epi, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), o.data_ptr(), N,
                                               M, N, K, 0, 64, 0.125, stream(dev)), "gemm")
                    outs.append(o)
                xd = x0.to(dev)
                _lib.check(lib.aaclip_gemm(code, _lib.EPI_BIAS_RESID, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(),
                                           xd.data_ptr(), N, M, N, K, 0, 0, 1.0, stream(dev)), "gemm")
                outs.append(xd)
                res[v] = outs
        finally:
            _variant(lib, 0)
        for a, b in zip(res[80], res[81]):
            assert torch.equal(a, b), f"{dtype} {(M, N, K)}: the half-tile kernel differs from the 8-wave kernel"
        ref = x0.double() + A.double().cpu() @ W.double().cpu().t() + bias.double().cpu()
        assert_close(res[81][2], ref, 2e-2 if dtype == "bf16" else 4e-3, 1e-2, f"half-tile resid {dtype}")


@pytest.mark.parametrize("B", [4, 16])
@pytest.mark.parametrize("exact", [False, True])
def test_blocks_half_tile_kernel_is_bit_identical(dev, exact, B):
    """Two full-size blocks through aaclip_blocks (B = 4: M = 5480 rows, the 256-row-tile regime; B = 16: 344 to 1376
    tiles per product, so the walking kernel takes several tiles per workgroup) in fp16x2, with the weights as drawn (4
    virtual tiles per K pair) and rounded through fp16 (3 tiles; the QKV product writes the attention kernel's e4m3 records
    in both): the stream after the blocks is bit-identical under all three GEMM kernels."""
    from model.clip import create_model
    lib = _lib.load()
    cfg = synth.ClipCfg()
    sd = synth.synth_clip_state_dict(cfg, 111)
    if exact:
        sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
    clip = create_model("ViT-L-14-336", 518, pretrained=None, precision="fp16x2", force_image_size=518)
    clip.load_state_dict(sd, strict=True)
    clip = clip.to(dev).eval()
    L = 1370
    x0 = synth.randn("t.hb.x", (B * L, 1024), 1.0, 5).to(dev)
    blocks = list(clip.visual.transformer.resblocks)[:2]
    res = {}
    try:
        for v in (80, 81, 82):
            _variant(lib, v)
            x = x0.clone()
            with torch.no_grad():
                engine.run_blocks(x, blocks, B, L, 16, F16X2)
            res[v] = x
    finally:
        _variant(lib, 0)
    assert torch.isfinite(res[80]).all() and float((res[80] - x0).abs().max()) > 0.1
    assert torch.equal(res[80], res[81])
    assert torch.equal(res[80], res[82])
