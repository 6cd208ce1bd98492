"""BASELINE.json configurations at size, and the benchmarked kernel regime against the reference's own numbers.

tests/golden/full4.npz is a full-size B = 4 record produced by running the REFERENCE (make_golden_full4.py).
B = 4 gives M = 5480 rows >= 4096: the large-batch kernels (256x256-tile GEMMs, LayerNorm passes folded into
the QKV / c_fc products, `aaclip_blocks_to` taps) that bench.py times at B = 64 -- full.npz (B = 2) only reaches
the small-batch kernels.  The B = 64 / B = 128 cases then tie the batch the benchmark runs to the same numbers
through bit-identity of an image's result across batches of the same kernel regime.

Tolerances.  North star (BASELINE.json): |a - b| <= 1e-3 + 1e-2 |b| against the fp32 reference.
  * exact-fp32 MFMA path: asserted at 1e-4 + 1e-3 |b| on every output INCLUDING the x100-amplified maps
    (measured, MI355X: max |err| 1.3e-5 on the 4-level map sum, 5e-7 on unit features).
  * fp16 MFMA path: north star asserted on every feature-level output (unit seg tokens: measured max |err|
    2.2e-4 = 0.16 of the bound; det token; pooled embedding).  The pre-blur maps are 50 x (cos_abnormal -
    cos_normal) per level, summed over 4 levels: the fp16 TOWER's rounding noise (10-bit mantissa on every
    GEMM operand, 24 layers) reaches them amplified -- measured max |err| 2.9e-3 per level and 4.6e-3 on the sum
    (rms 1.2e-3; |ref| up to 5.3), i.e. up to 2.8x the north-star bound on 0.9 % of the pixels.  Running ln_post /
    seg_proj / normalise / the anchor dot in exact fp32 or fp64 on the SAME fp16 tower output changes nothing
    (tools/map_error_probe.py: 4.44e-3 vs 4.65e-3), so the split-precision head SURVEY section 7 suggested cannot
    close it; only more mantissa bits in the tower can, which is what precision='fp32' is.  Asserted here at the
    measured level with headroom: 4e-3 + 1e-2 |b| per level, 6e-3 + 1e-2 |b| on the level sum, and every
    comparison writes its actual maximum error to gpurun_out/parity_errors.json (committed as
    profiles/r02_parity_errors.json).  AUROC parity (|delta| <= 1e-3) is asserted in test_gpu_parity.py.
"""
import json
import os

import numpy as np
import pytest
import torch

from aaclip_hip import engine, synth
from aaclip_hip._lib import BF16, F16, F16X2, F32
from conftest import GOLDEN, PARITY_ERRORS, REPO

pytestmark = pytest.mark.gpu
T = torch.from_numpy
NAME = {F32: "fp32", F16: "fp16", BF16: "bf16", F16X2: "fp16x2"}
NORTH_STAR = (1e-3, 1e-2)                                      # BASELINE.json: 1e-3 abs + 1e-2 rel vs the fp32 reference
# fp16x2 (split fp16 on the 16-bit MFMAs) is held to the north star on EVERY output: features, raw taps, pooled
# embedding, per-level and summed pre-blur maps.  Plain fp16 meets it on features only (see the module docstring).
FEATURE_TOL = {F32: (1e-4, 1e-3), F16: NORTH_STAR, F16X2: NORTH_STAR}
MAP_TOL = {F32: ((1e-4, 1e-3), (1e-4, 1e-3)), F16: ((4e-3, 1e-2), (6e-3, 1e-2)),   # (per level, level sum)
           F16X2: (NORTH_STAR, NORTH_STAR)}
TAP_TOL = {F32: (1e-4, 1e-3), F16: (4e-3, 1e-2), F16X2: NORTH_STAR}   # raw residual stream, values O(1..4)

ERRORS = PARITY_ERRORS      # shared with the other GPU test files, dumped once per session (tests/conftest.py)


def compare(name, a, b, atol, rtol):
    """assert |a-b| <= atol + rtol|b| and record the actual numbers (max abs error, its share of the north-star
    bound 1e-3 + 1e-2|b|, and of the asserted bound)."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert torch.isfinite(a).all(), f"{name}: non-finite output"
    err = (a - b).abs()
    ERRORS[name] = {
        "max_abs_err": float(err.max()), "rms_err": float(err.pow(2).mean().sqrt()),
        "max_ratio_to_north_star": float((err / (1e-3 + 1e-2 * b.abs())).max()),
        "frac_outside_north_star": float((err > 1e-3 + 1e-2 * b.abs()).double().mean()),
        "max_ratio_to_asserted": float((err / (atol + rtol * b.abs())).max()),
        "asserted": [atol, rtol], "ref_abs_max": float(b.abs().max()), "n": int(b.numel()),
    }
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), (f"{name}: {int(bad.sum())}/{bad.numel()} outside {atol}+{rtol}*|ref|; "
                           f"max err {err.max().item():.3e}")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g4():
    return np.load(os.path.join(GOLDEN, "full4.npz"))


@pytest.fixture(scope="module")
def weights():
    cfg = synth.ClipCfg()
    return (cfg, synth.synth_clip_state_dict(cfg, 111), synth.synth_image_adapter_state_dict(cfg, seed=111),
            synth.synth_text_adapter_state_dict(cfg, seed=111))


def build(dev, precision, weights):
    from model.clip import create_model
    from model.adapter import AdaptedCLIP
    cfg, sd, ia, ta = weights
    clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=precision, force_image_size=518)
    clip.load_state_dict(sd, strict=True)
    model = AdaptedCLIP(clip, relu=False)
    model.image_adapter.load_state_dict(ia, strict=True)
    model.text_adapter.load_state_dict(ta, strict=True)
    return model.to(dev).eval()


def sampled(g, name, t):
    assert tuple(g[f"{name}.shape"]) == tuple(t.shape), (name, tuple(t.shape))
    return t.detach().reshape(-1).cpu()[T(g[f"{name}.idx"])], T(g[f"{name}.val"])


def images4(g4):
    return synth.synth_images(4, 518, seed=int(g4["full4.seed"]))


@pytest.mark.parametrize("code", [F32, F16X2, F16])
def test_full_b4_adapted_forward_vs_reference_golden(dev, g4, weights, code):
    """BASELINE config 3's forward on the large-batch kernels vs the reference's B = 4 numbers: seg tokens,
    det token, pre-blur map of every level and the level sum (anchors: the reference's own 'bottle' anchors)."""
    model = build(dev, NAME[code], weights)
    tag = NAME[code]
    with torch.no_grad():
        seg, det, _ = model(images4(g4).to(dev))
    atol, rtol = FEATURE_TOL[code]
    for i in range(4):
        a, b = sampled(g4, f"full4.seg{i}", seg[i])
        compare(f"{tag}.b4.seg{i}", a, b, atol, rtol)
        assert float((seg[i].norm(dim=-1) - 1).abs().max()) < 1e-5
    compare(f"{tag}.b4.det", det, T(g4["full4.det"]), atol, rtol)
    anchors = T(np.load(os.path.join(GOLDEN, "full.npz"))["full.anchors_bottle"]).to(dev)
    (la, lr), (sa, sr) = MAP_TOL[code]
    total = 0
    for i in range(4):
        raw = engine.anomaly_map([seg[i]], anchors, 37, 1, 1.0)          # S == grid, ksize 1: the pre-blur map
        compare(f"{tag}.b4.map_pre_blur{i}", raw, T(g4[f"full4.map_pre_blur{i}"]), la, lr)
        total = total + raw
    fused = engine.anomaly_map(list(seg), anchors, 37, 1, 1.0)            # all levels in one launch
    compare(f"{tag}.b4.map_pre_blur_sum", fused, T(g4["full4.map_pre_blur_sum"]), sa, sr)
    compare(f"{tag}.b4.map_fused_vs_level_sum", fused, total, 1e-5, 1e-6)


@pytest.mark.parametrize("code", [F32, F16X2, F16])
def test_full_b4_fp16_native_weights(dev, weights, code):
    """The deployment case: CLIP weights that are exactly representable in fp16 (OpenAI's checkpoint is stored in fp16
    and the reference widens it on load, model/clip.py:88), so the fp16 path's weight conversion is lossless and only
    the ACTIVATION rounding of the tower is left.  Golden: the reference run with the seeded CLIP weights rounded
    through fp16 (tests/golden/make_golden_full4h.py).  Asserted at the same bounds as the fp32-weight record; the
    actual errors go to parity_errors.json next to it (`*.b4h.*`) -- the measured difference between the two records
    is the share of the map error that came from rounding fp32 weights, which real checkpoints do not have."""
    g = np.load(os.path.join(GOLDEN, "full4h.npz"))
    cfg, sd, ia, ta = weights
    sdh = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
    model = build(dev, NAME[code], (cfg, sdh, ia, ta))
    tag = NAME[code]
    with torch.no_grad():
        seg, det, _ = model(synth.synth_images(4, 518, seed=int(g["full4h.seed"])).to(dev))
    atol, rtol = FEATURE_TOL[code]
    for i in range(4):
        a_, b_ = sampled(g, f"full4h.seg{i}", seg[i])
        compare(f"{tag}.b4h.seg{i}", a_, b_, atol, rtol)
    compare(f"{tag}.b4h.det", det, T(g["full4h.det"]), atol, rtol)
    anchors = T(np.load(os.path.join(GOLDEN, "full.npz"))["full.anchors_bottle"]).to(dev)
    (la, lr), (sa, sr) = MAP_TOL[code]
    for i in range(4):
        raw = engine.anomaly_map([seg[i]], anchors, 37, 1, 1.0)
        compare(f"{tag}.b4h.map_pre_blur{i}", raw, T(g[f"full4h.map_pre_blur{i}"]), la, lr)
    fused = engine.anomaly_map(list(seg), anchors, 37, 1, 1.0)
    compare(f"{tag}.b4h.map_pre_blur_sum", fused, T(g["full4h.map_pre_blur_sum"]), sa, sr)


@pytest.mark.parametrize("code", [F32, F16X2, F16])
def test_full_b4_encode_image_taps_vs_reference_golden(dev, g4, weights, code):
    """BASELINE config 2 as written -- CLIP.encode_image(image, [6, 12, 18, 24]) -- on the large-batch kernels
    (256-tile GEMMs, ln folds, aaclip_blocks_to taps without copies) vs the reference's B = 4 numbers."""
    model = build(dev, NAME[code], weights)
    tag = NAME[code]
    with torch.no_grad():
        pooled, taps = model.clipmodel.encode_image(images4(g4).to(dev), [6, 12, 18, 24])
    assert len(taps) == 4
    atol, rtol = TAP_TOL[code]
    for k, t in zip((6, 12, 18, 24), taps):
        a, b = sampled(g4, f"full4.tap{k}", t)
        compare(f"{tag}.b4.tap{k}", a, b, atol, rtol)
    compare(f"{tag}.b4.pooled", pooled, T(g4["full4.pooled"]), *TAP_TOL[code])


# plain fp16 on the outlier record: recorded, asserted only at a loose measured level (it is outside the north star on
# the plain record already); (per-level / summed maps, taps + pooled)
OUTLIER_FP16_TOL = (MAP_TOL[F16][0], MAP_TOL[F16][1], TAP_TOL[F16])   # measured: maps 8e-4 / 1.7e-3, taps 2.3e-3 (1.6x the north star)


@pytest.mark.parametrize("code", [F32, F16X2, F16])
def test_full_b4_outlier_weights(dev, weights, code):
    """The record with OUTLIER channels (tests/golden/make_golden_full4o.py: the reference run on
    synth.outlier_edit weights -- residual-stream channels at +-80 ... +-600 from block 6 on, LayerNorm gains x5..x10,
    GELU outputs up to 2500, i.e. beyond the +-448 of the e4m3 correction planes).  Every other parity record is
    Gaussian-initialised weights; this is the one that looks like a trained checkpoint.  The north star is asserted for
    the exact-fp32 mode AND for fp16x2 on raw taps, pooled embedding, unit seg tokens, det token, per-level and summed
    pre-blur maps."""
    g = np.load(os.path.join(GOLDEN, "full4o.npz"))
    assert float(g["full4o.stream_absmax"].min()) > 500 and float(g["full4o.gelu_absmax"].max()) > 2000   # it IS an outlier case
    cfg, sd, ia, ta = weights
    model = build(dev, NAME[code], (cfg, synth.outlier_edit(sd, cfg, 111), ia, ta))
    tag = NAME[code] + ".b4o"
    img = synth.synth_images(4, 518, seed=int(g["full4o.seed"])).to(dev)
    with torch.no_grad():
        seg, det, _ = model(img)
        pooled, taps = model.clipmodel.encode_image(img, [6, 12, 18, 24])
    if code == F16:
        (la, lr), (sa, sr), tap_tol = OUTLIER_FP16_TOL
        feat_tol = tap_tol
    else:
        (la, lr), (sa, sr), tap_tol, feat_tol = NORTH_STAR, NORTH_STAR, NORTH_STAR, NORTH_STAR
    for k, t in zip((6, 12, 18, 24), taps):
        a, b = sampled(g, f"full4o.tap{k}", t)
        assert float(b.abs().max()) > 500
        compare(f"{tag}.tap{k}", a, b, *tap_tol)
    compare(f"{tag}.pooled", pooled, T(g["full4o.pooled"]), *tap_tol)
    for i in range(4):
        a, b = sampled(g, f"full4o.seg{i}", seg[i])
        compare(f"{tag}.seg{i}", a, b, *feat_tol)
    compare(f"{tag}.det", det, T(g["full4o.det"]), *feat_tol)
    anchors = T(np.load(os.path.join(GOLDEN, "full.npz"))["full.anchors_bottle"]).to(dev)
    for i in range(4):
        raw = engine.anomaly_map([seg[i]], anchors, 37, 1, 1.0)
        compare(f"{tag}.map_pre_blur{i}", raw, T(g[f"full4o.map_pre_blur{i}"]), la, lr)
    fused = engine.anomaly_map(list(seg), anchors, 37, 1, 1.0)
    compare(f"{tag}.map_pre_blur_sum", fused, T(g["full4o.map_pre_blur_sum"]), sa, sr)


@pytest.mark.parametrize("code", [F16X2, F16])
def test_config2_b64_encode_image_is_the_b4_result(dev, g4, weights, code):
    """BASELINE config 2 at its size: encode_image(img64, [6,12,18,24]), in the benchmarked mode (fp16x2) and in plain
    fp16.  Images 0..3 are the golden B = 4 images; every tap row and pooled row of theirs must be BIT-IDENTICAL to
    the B = 4 run (same kernels, rows are independent), which test_full_b4_encode_image_taps_vs_reference_golden pins
    to the reference.  Two runs are bit-identical; all 64 results are finite and differ between images."""
    model = build(dev, NAME[code], weights)
    img4 = images4(g4)
    img64 = torch.cat([img4, synth.synth_images(60, 518, seed=64)], dim=0).to(dev)
    with torch.no_grad():
        p4, t4 = model.clipmodel.encode_image(img64[:4], [6, 12, 18, 24])
        p64, t64 = model.clipmodel.encode_image(img64, [6, 12, 18, 24])
        p64b, t64b = model.clipmodel.encode_image(img64, [6, 12, 18, 24])
    assert p64.shape == (64, 768) and all(t.shape == (64, 1370, 1024) for t in t64)
    assert torch.equal(p64[:4], p4) and torch.equal(p64, p64b)
    for a, b, c in zip(t64, t4, t64b):
        assert torch.equal(a[:4], b)
        assert torch.equal(a, c)
        assert torch.isfinite(a).all()
    assert float((p64[5] - p64[6]).abs().max()) > 1e-3
    a, b = sampled(g4, "full4.tap24", t64[3][:4])
    compare(f"{NAME[code]}.b64.tap24_first4", a, b, *TAP_TOL[code])


@pytest.mark.parametrize("code", [F16X2, F16])
def test_config3_b64_full_path_is_the_b4_result(dev, g4, weights, code):
    """BASELINE config 3 at its size: AdaptedCLIP.forward + fused anomaly map at B = 64 (fp16x2 and fp16); images 0..3
    are bit-identical to the B = 4 run that is pinned to the reference golden."""
    import forward_utils as FU
    model = build(dev, NAME[code], weights)
    img64 = torch.cat([images4(g4), synth.synth_images(60, 518, seed=64)], dim=0).to(dev)
    anchors = T(np.load(os.path.join(GOLDEN, "full.npz"))["full.anchors_bottle"]).to(dev)
    with torch.no_grad():
        seg4, det4, _ = model(img64[:4])
        seg, det, _ = model(img64)
        m4 = FU.calculate_anomaly_map(seg4, anchors, 518, domain="Industrial")
        m64 = FU.calculate_anomaly_map(seg, anchors, 518, domain="Industrial")
    for a, b in zip(seg, seg4):
        assert torch.equal(a[:4], b)
    assert torch.equal(det[:4], det4) and torch.equal(m64[:4], m4)
    assert m64.shape == (64, 518, 518) and torch.isfinite(m64).all()
    a, b = sampled(g4, "full4.seg3", seg[3][:4])
    compare(f"{NAME[code]}.b64.seg3_first4", a, b, *FEATURE_TOL[code])


@pytest.mark.parametrize("code", [F16X2, F16, BF16])
def test_config5_b128_four_tap_layers(dev, g4, weights, code):
    """BASELINE config 5: 16-bit MFMA path, batch 128 per GPU, multi-layer patch-feature extraction (4 tap layers).
    Size-independent properties: images 0..3 bit-identical to the B = 4 run of the same dtype, run-to-run
    determinism, finite distinct results; fp16 additionally pinned to the reference through the B = 4 golden."""
    model = build(dev, NAME[code], weights)
    img = torch.cat([images4(g4), synth.synth_images(124, 518, seed=128)], dim=0).to(dev)
    with torch.no_grad():
        p4, t4 = model.clipmodel.encode_image(img[:4], [6, 12, 18, 24])
        p, t = model.clipmodel.encode_image(img, [6, 12, 18, 24])
        seg, det, _ = model(img)
        seg4, det4, _ = model(img[:4])
    assert p.shape == (128, 768) and all(x.shape == (128, 1370, 1024) for x in t)
    assert torch.equal(p[:4], p4)
    for a, b in zip(t, t4):
        assert torch.equal(a[:4], b) and torch.isfinite(a).all()
    for a, b in zip(seg, seg4):
        assert a.shape == (128, 1369, 768) and torch.equal(a[:4], b)
    assert torch.equal(det[:4], det4)
    assert float((p[100] - p[101]).abs().max()) > 1e-3
    if code in (F16, F16X2):
        a, b = sampled(g4, "full4.tap18", t[2][:4])
        compare(f"{NAME[code]}.b128.tap18_first4", a, b, *TAP_TOL[code])
    else:   # bf16: 8-bit mantissa, offered, not the parity path
        a, b = sampled(g4, "full4.seg3", seg[3][:4])
        compare("bf16.b128.seg3_first4", a, b, 1e-2, 5e-2)


@pytest.mark.parametrize("seed", [1017, 4242])
def test_fp16x2_keeps_the_north_star_on_another_seed(dev, seed):
    """The golden record is ONE set of weights and images.  A second one (other synthetic weights, fp16-exact like
    OpenAI's, other images; no reference numbers exist for it: "parity unpinned", the exact-fp32 mode of this build --
    0.006 of the bound on the golden record -- stands in): fp16x2 must keep taps, pooled embedding, seg / det tokens and
    per-level pre-blur maps inside 1e-3 + 1e-2 |ref| there as well (measured over six seeds: 0.44-0.48 of the bound,
    tools/margin_probe.py; the second seed promotes that probe into the suite)."""
    import forward_utils as FU
    cfg = synth.ClipCfg()
    sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in synth.synth_clip_state_dict(cfg, seed).items()}
    w = (cfg, sd, synth.synth_image_adapter_state_dict(cfg, seed=seed), synth.synth_text_adapter_state_dict(cfg, seed=seed))
    img = synth.synth_images(4, 518, seed=seed).to(dev)
    anchors = torch.nn.functional.normalize(torch.randn(768, 2, generator=torch.Generator().manual_seed(seed)), dim=0).to(dev)
    outs = {}
    for code in (F32, F16X2):
        model = build(dev, NAME[code], w)
        with torch.no_grad():
            pooled, taps = model.clipmodel.encode_image(img, [6, 12, 18, 24])
            seg, det, _ = model(img)
            maps = [FU.calculate_similarity_map(s, anchors, 37)[:, 1] for s in seg]
        outs[code] = [pooled.float().cpu(), det.float().cpu()] + [t.float().cpu() for t in taps] + \
                     [s.float().cpu() for s in seg] + [m.float().cpu() for m in maps]
        del model
        torch.cuda.empty_cache()
    names = ["pooled", "det"] + [f"tap{6 * (k + 1)}" for k in range(4)] + [f"seg{k}" for k in range(4)] + \
            [f"map_pre_blur{k}" for k in range(4)]
    for name, a, b in zip(names, outs[F16X2], outs[F32]):
        compare(f"fp16x2.seed{seed}.{name}", a, b, *NORTH_STAR)
