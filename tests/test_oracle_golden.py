"""Pin the CPU oracle to golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from aaclip_hip import synth
from oracle import aaclip_oracle as O

T = torch.from_numpy


def close(a, b, atol=2e-5, rtol=2e-4):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}, worst excess {(err - tol).max().item():.3e}"


def sampled(golden, name, t, atol=2e-5, rtol=2e-4):
    assert tuple(golden[f"{name}.shape"]) == tuple(t.shape)
    f = t.reshape(-1)
    close(f[T(golden[f"{name}.idx"])], golden[f"{name}.val"], atol, rtol)
    s = golden[f"{name}.abssum"]
    assert abs(f.double().abs().sum().item() - s) <= 1e-4 * s


@pytest.fixture(scope="module")
def tiny():
    cfg = synth.tiny_cfg()
    return cfg, synth.synth_clip_state_dict(cfg, seed=7), synth.synth_images(3, cfg.image_size, seed=7)


def test_tiny_encode_image(golden_tiny, tiny):
    cfg, sd, img = tiny
    pooled, taps = O.encode_image(img, sd, cfg.vision.heads, [1, 3])
    close(pooled, golden_tiny["tiny.pooled"])
    close(taps[0], golden_tiny["tiny.tap1"])
    close(taps[1], golden_tiny["tiny.tap3"])


def test_tiny_encode_text(golden_tiny, tiny):
    cfg, sd, _ = tiny
    tok = T(golden_tiny["tiny.tokens"])
    close(O.encode_text(tok, sd, cfg.text.heads), golden_tiny["tiny.text"])


def test_tiny_adapted_stream(golden_tiny, tiny):
    cfg, sd, img = tiny
    ia = synth.synth_image_adapter_state_dict(cfg, until=2, levels=2, seed=7)
    _, _, stream = O.adapted_visual_forward(img, sd, ia, cfg.vision.heads, image_adapt_until=2,
                                            levels=(2, 3), return_stream=True)
    close(stream[-1], golden_tiny["tiny.adapted_stream"])


def test_resize_pos_embed(golden_tiny):
    close(O.resize_pos_embed(T(golden_tiny["resize.in"]), 37), golden_tiny["resize.out"], atol=1e-6)


def test_similarity_map_pieces(golden_tiny):
    pf, tf = T(golden_tiny["map.pf"]), T(golden_tiny["map.tf"])
    close(O.similarity_map(pf, tf, 70, test=False), golden_tiny["map.train"], atol=1e-5)
    s = 100.0 * torch.matmul(pf, tf)
    pre = ((s[..., 1] + 1 - s[..., 0]) / 2).view(2, 1, 5, 5)
    close(pre, golden_tiny["map.pre_blur"], atol=1e-5)
    close(O.bilinear_align_corners(pre, 70), golden_tiny["map.pre_blur_up"], atol=1e-5)


def test_blur_properties():
    # parity unpinned (kornia absent): check the restated algorithm's invariants
    x = torch.ones(1, 1, 9, 9)
    assert torch.allclose(O.gaussian_blur2d(x, 7, 1.0), x, atol=1e-6)
    k = O.gaussian_kernel1d(7, 1.0, torch.float64)
    assert abs(k.sum().item() - 1) < 1e-12 and torch.allclose(k, k.flip(0))
    ref = torch.exp(-torch.arange(-3, 4, dtype=torch.float64) ** 2 / 2)
    assert torch.allclose(k, ref / ref.sum())
    d = torch.zeros(1, 1, 11, 11)
    d[0, 0, 5, 5] = 1
    b = O.gaussian_blur2d(d, 7, 1.0)
    assert torch.allclose(b[0, 0, 2:9, 2:9], torch.outer(k, k).float(), atol=1e-7)


@pytest.fixture(scope="module")
def full():
    cfg = synth.ClipCfg()
    return (cfg, synth.synth_clip_state_dict(cfg, seed=111), synth.synth_image_adapter_state_dict(cfg, seed=111),
            synth.synth_text_adapter_state_dict(cfg, seed=111))


def test_full_adapted_visual(golden_full, full):
    cfg, sd, ia, _ = full
    torch.set_num_threads(8)
    img = synth.synth_images(2, 518, seed=111)
    seg, det, stream = O.adapted_visual_forward(img, sd, ia, cfg.vision.heads, return_stream=True)
    for i in range(4):
        sampled(golden_full, f"full.seg{i}", seg[i], atol=2e-5, rtol=1e-3)
        sampled(golden_full, f"full.stream{i}", stream[i][:, 1:], atol=2e-4, rtol=1e-3)
    close(det, golden_full["full.det"], atol=1e-5, rtol=1e-3)
    # map pieces and the reference's image-score quirk on the golden anchors
    anchors = T(golden_full["full.anchors_bottle"])
    tfb = anchors.unsqueeze(0).repeat(2, 1, 1)
    for i in range(4):
        s = 100.0 * torch.matmul(seg[i], tfb)
        pre = ((s[..., 1] + 1 - s[..., 0]) / 2).view(2, 37, 37)
        close(pre, golden_full[f"full.map_pre_blur{i}"], atol=2e-3, rtol=1e-3)
    sampled(golden_full, "full.map_train3", O.similarity_map(seg[3], tfb, 518, test=False), atol=1e-4, rtol=1e-3)
    close(O.image_score_reference_quirk(det, tfb), golden_full["full.image_pred_quirk"], atol=1e-5)


def test_full_text(golden_full, full):
    cfg, sd, _, ta = full
    tok = T(golden_full["full.text_tokens"])
    close(O.adapted_encode_text(tok, sd, ta, cfg.text.heads), golden_full["full.text_adapted"], atol=2e-5, rtol=1e-3)
    close(O.encode_text(tok, sd, cfg.text.heads), golden_full["full.text_plain"], atol=2e-5, rtol=1e-3)


def test_full_encode_image(golden_full, full):
    cfg, sd, _, _ = full
    img = synth.synth_images(1, 518, seed=111)
    pooled, taps = O.encode_image(img, sd, cfg.vision.heads, [6, 24])
    close(pooled, golden_full["full.pooled"], atol=1e-4, rtol=1e-3)
    sampled(golden_full, "full.tap6", taps[0], atol=2e-4, rtol=1e-3)
    sampled(golden_full, "full.tap24", taps[1], atol=2e-4, rtol=1e-3)


def test_outlier_record_pins_the_oracle_too():
    """tests/golden/full4o.npz: the reference on synth.outlier_edit weights (residual-stream channels at +-600, GELU
    outputs at 2500).  The oracle's fp32 restatement must reproduce the reference there as well (first two images:
    rows are independent, and the record samples the flattened [4, 1370, 1024] taps)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "full4o.npz"))
    cfg = synth.ClipCfg()
    sd = synth.outlier_edit(synth.synth_clip_state_dict(cfg, 111), cfg, 111)
    # the edit is what its docstring says
    assert float(sd["visual.transformer.resblocks.5.mlp.c_proj.bias"].abs().max()) > 590
    assert float(sd["visual.transformer.resblocks.15.mlp.c_fc.bias"].max()) == 2500.0
    img = synth.synth_images(2, 518, seed=int(g["full4o.seed"]))
    torch.set_num_threads(8)
    with torch.no_grad():
        pooled, taps = O.encode_image(img, sd, cfg.vision.heads, [6, 24])
    close(pooled, g["full4o.pooled"][:2], atol=1e-4, rtol=1e-3)
    for k, t in zip((6, 24), taps):
        idx, val = T(g[f"full4o.tap{k}.idx"]), T(g[f"full4o.tap{k}.val"])
        keep = idx < t.numel()
        assert float(val[keep].abs().max()) > 500
        close(t.reshape(-1)[idx[keep]], val[keep], atol=2e-4, rtol=1e-3)


# ---------------------------------------------------------------------------------------------
# "CLIP surgery" tap path (reference transformer.py:102-152,406-425), golden from the reference
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_surgery():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "surgery.npz"))


@pytest.mark.parametrize("B", [3, 1])
@pytest.mark.parametrize("dpam", [2, 3])
def test_surgery_encode_image_matches_reference(golden_surgery, B, dpam):
    cfg = synth.tiny_cfg()
    sd = synth.synth_clip_state_dict(cfg, seed=7)
    img = synth.synth_images(B, cfg.image_size, seed=7)
    pooled, taps = O.encode_image(img, sd, cfg.vision.heads, [1, 2, 3], dpam_layer=dpam)
    g = golden_surgery
    assert torch.allclose(pooled, torch.from_numpy(g[f"b{B}.dpam{dpam}.pooled"]), atol=2e-5, rtol=1e-5)
    for i, t in enumerate(taps):
        assert torch.allclose(t, torch.from_numpy(g[f"b{B}.dpam{dpam}.tap{i + 1}"]), atol=2e-5, rtol=1e-5), i


def test_surgery_depends_on_batch_composition(golden_surgery):
    """The reference quirk the restatement reproduces: attention over the batch axis."""
    g = golden_surgery
    assert not np.allclose(g["b3.dpam3.tap3"][:1], g["b1.dpam3.tap3"], atol=1e-3)
    assert np.allclose(g["b3.dpam3.tap1"][:1], g["b1.dpam3.tap1"], atol=1e-5)      # block 1 is untouched


def test_iqm_branch_vs_reference_golden():
    """the oracle's IQM branch (model/adapter.py:186-269 + model/iqm.py, eval mode) and the IQM
    anomaly maps of test_last.py:102-138 against what the REFERENCE computed with the same seeded weights
    (tests/golden/make_golden_iqm.py), full-size model, B = 4."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "iqm.npz"))
    g4 = np.load(os.path.join(GOLDEN, "full4.npz"))
    anchors = T(np.load(os.path.join(GOLDEN, "full.npz"))["full.anchors_bottle"])
    cfg = synth.ClipCfg()
    sd = synth.synth_clip_state_dict(cfg, 111)
    ia = synth.synth_image_adapter_state_dict(cfg, seed=111)
    isd = synth.synth_iqm_state_dict(cfg, seed=int(g["iqm.seed"]))
    img = synth.synth_images(4, 518, seed=int(g4["full4.seed"]))
    torch.set_num_threads(8)
    with torch.no_grad():
        seg, det, h = O.adapted_visual_forward_iqm(img, sd, ia, isd, anchors.unsqueeze(0).repeat(4, 1, 1), cfg.vision.heads)
        total = O.iqm_anomaly_map(seg, h, 518)
    ref = T(g["iqm.last_hidden_state"])
    assert h.shape == ref.shape == (4, 2, 768)
    assert float((h - ref).abs().max()) < 2e-4, float((h - ref).abs().max())      # values O(1..4), fp32 both sides
    f = total.reshape(-1)
    assert tuple(g["iqm.map_sum.shape"]) == tuple(total.shape)
    assert float((f[T(g["iqm.map_sum.idx"])] - T(g["iqm.map_sum.val"])).abs().max()) < 2e-5
    assert float((total[0][::7, ::7] - T(g["iqm.map_sum_full0"])).abs().max()) < 2e-5
