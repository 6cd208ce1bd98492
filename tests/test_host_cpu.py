"""CPU-side checks: the C-ABI library loads and exports every declared symbol,
host logic (tokenizer, prompts, state-dict contract, factory errors) and that
the product path refuses to run without a GPU instead of falling back."""
import ctypes
import json
import os
import re

import pytest
import torch

from aaclip_hip import _lib, engine, synth
from conftest import GOLDEN, REPO


def _header_abi_version():
    header = open(os.path.join(REPO, "include", "aaclip.h")).read()
    return int(re.search(r"#define\s+AACLIP_ABI_VERSION\s+(\d+)", header).group(1))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    assert lib.aaclip_version() == _header_abi_version() == _lib.ABI_VERSION
    assert lib.aaclip_is_measurement_build() == 0      # the product library carries no ablation / stamp kernels
    header = open(os.path.join(REPO, "include", "aaclip.h")).read()
    declared = set(re.findall(r"\b(aaclip_[a-z_0-9]+)\s*\(", header))
    declared.discard("aaclip_block_weights")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_graft_entry_build_runs():
    """The driver's build check: make + load + ABI version from the header (it once asserted a stale constant)."""
    import __graft_entry__ as G
    G.build()


def test_block_weights_struct_matches_header():
    header = open(os.path.join(REPO, "include", "aaclip.h")).read()
    body = re.search(r"typedef struct aaclip_block_weights \{(.*?)\} aaclip_block_weights;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(\w+)\s*;", body)
    assert fields == [n for n, _ in _lib.BlockWeights._fields_]
    assert _lib.BlockWeights().struct_bytes == ctypes.sizeof(_lib.BlockWeights) == 8 * len(fields)


def test_short_or_old_block_struct_is_rejected():
    """A binding generated from an older header (13 pointer fields: ABI 1; 19: ABI 2; neither has the size field)
    must get rc < 0 from every entry point that takes the struct -- before anything is read past its end or
    launched.  Runs without a GPU: the check precedes every HIP call."""
    lib = _lib.load()

    class Old13(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in (
            "ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc_w", "fc_b", "proj_w",
            "proj_b", "adapter_w")]

    class Old19(ctypes.Structure):
        _fields_ = Old13._fields_ + [(n, ctypes.c_void_p) for n in (
            "fc_w_fold", "fc_fold_s", "fc_fold_b", "qkv_w_fold", "qkv_fold_s", "qkv_fold_b")]

    block = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, *([ctypes.c_int] * 7),
                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)(("aaclip_block", lib))
    blocks_to = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.c_float, *([ctypes.c_int] * 7), ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_void_p)(("aaclip_blocks_to", lib))
    for Old in (Old13, Old19):
        w = Old()
        for n, _ in Old._fields_:
            setattr(w, n, 0x7f0000001000)      # plausible device addresses: nothing here may be dereferenced
        rc = block(0x7f0000002000, ctypes.addressof(w), 0.1, 4, 1370, 1024, 16, 4096, 0, _lib.F16, 0x7f0000003000,
                   1 << 40, None)
        assert rc < 0 and b"struct_bytes" in lib.aaclip_last_error()
        rc = blocks_to(0x7f0000002000, 0x7f0000004000, ctypes.addressof(w), 1, 0.1, 4, 1370, 1024, 16, 4096, 0,
                       _lib.F16, 0x7f0000003000, 1 << 40, None)
        assert rc < 0 and b"struct_bytes" in lib.aaclip_last_error()
    # a current struct with a tampered size is refused too; the right size passes this check (and then fails the
    # next one, a null weight pointer, still without touching the GPU)
    w = _lib.BlockWeights()
    w.struct_bytes = 13 * 8
    rc = lib.aaclip_block(0x7f0000002000, ctypes.byref(w), 0.1, 4, 1370, 1024, 16, 4096, 0, _lib.F16, 0x7f0000003000,
                          1 << 40, None)
    assert rc < 0 and b"struct_bytes" in lib.aaclip_last_error()


def test_product_library_refuses_measurement_variants():
    """Timing ablations (wrong results), A/B variants and stamp kernels are not in libaaclip_hip.so: selecting one is
    an error and leaves the selection unchanged; no environment variable selects kernels any more."""
    lib = _lib.load()
    for v in (2, 5, 11, 17, 20, 46, 70, 2 << 8, 1 << 18, -1):
        assert lib.aaclip_set_gemm_variant(v) < 0, v
    assert b"measure" in lib.aaclip_last_error() or b"unknown" in lib.aaclip_last_error()
    for v in (1, 1 << 8, 1 << 16, 1 << 17, 0):
        assert lib.aaclip_set_gemm_variant(v) == 0, v
    assert lib.aaclip_debug_gemm_stamps(None, 4) < 0
    src = open(os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "_lib.py")).read()
    assert "AACLIP_GEMM_VARIANT" not in src


def test_measurement_library_has_no_substitute_kernels():
    """Regression for the round-1 GPU fault: an fp32-storing timing ablation was launched for a
    product whose output buffer is 16-bit and overran it.  The ablations now live in libaaclip_hip_measure.so only,
    exist for the fp32-output epilogue only, and any other (variant, epilogue) pair returns rc < 0 WITHOUT launching
    (so this runs on a CPU box with made-up pointers)."""
    if not os.path.exists(_lib.MEASURE_LIB_PATH):
        pytest.skip("measurement library not built (make -C aa-clip-iqm_amd/csrc measure)")
    m = ctypes.CDLL(_lib.MEASURE_LIB_PATH)
    m.aaclip_last_error.restype = ctypes.c_char_p
    res, args = _lib.SIGNATURES["aaclip_gemm"]
    m.aaclip_gemm.restype, m.aaclip_gemm.argtypes = res, args
    assert m.aaclip_is_measurement_build() == 1 and m.aaclip_version() == _lib.ABI_VERSION
    M, N, K = 8192, 1024, 1024
    for variant in (4, 5, 11, 12, 13, 14, 15, 16, 17, 46):      # 32x32x16 ablations, 16x16x32 ablations + stamp build
        assert m.aaclip_set_gemm_variant(variant) == 0
        for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU, _lib.EPI_BIAS_RESID):
            rc = m.aaclip_gemm(_lib.F16, epi, 0x7f0000001000, K, 0x7f0000002000, 0x7f0000003000, 0x7f0000004000, N,
                               M, N, K, 0, 0, 1.0, None)
            assert rc < 0 and b"fp32-output epilogue only" in m.aaclip_last_error(), (variant, epi)
    assert m.aaclip_set_gemm_variant(61) < 0
    for av in (4, 5, 7):                                         # attention variants: 0-3 and 6 exist
        assert m.aaclip_set_gemm_variant(av << 8) < 0, av
    for av in (3, 6, 0):
        assert m.aaclip_set_gemm_variant(av << 8) == 0, av
    assert m.aaclip_set_gemm_variant(0) == 0


def test_workspace_bytes_monotone():
    lib = _lib.load()
    a = lib.aaclip_workspace_bytes(_lib.F16, 1370, 1024, 4096, 768)
    b = lib.aaclip_workspace_bytes(_lib.F16, 2 * 1370, 1024, 4096, 768)
    c = lib.aaclip_workspace_bytes(_lib.F32, 1370, 1024, 4096, 768)
    assert 0 < a < b and a < c
    assert a >= 1370 * (1024 * 2 + 4096 * 2)


def test_argument_errors_cross_the_abi_as_codes():
    lib = _lib.load()
    rc = lib.aaclip_layernorm(None, None, None, None, _lib.F32, 4, 1024, 1e-5, None)
    assert rc < 0 and b"null" in lib.aaclip_last_error()
    rc = lib.aaclip_gemm(_lib.F16, 0, 1, 64, 1, 1, 1, 100, 8, 100, 64, 0, 0, 1.0, None)
    assert rc < 0 and b"multiple of 128" in lib.aaclip_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "gemm")


def test_tokenizer_matches_reference_fixture():
    from model.tokenizer import tokenize
    table = json.load(open(os.path.join(GOLDEN, "token_ids.json")))
    sents = sorted(table)[:: max(1, len(table) // 40)]
    out = tokenize(sents)
    assert out.dtype == torch.int32 and out.shape == (len(sents), 77)
    for row, s in zip(out, sents):
        ids = table[s]
        assert row[: len(ids)].tolist() == ids and int(row[len(ids):].abs().sum()) == 0
        assert int(row.argmax()) == len(ids) - 1  # EOT is the largest id


def test_own_bpe_matches_reference_fixture_when_vocab_available():
    from model import tokenizer as T
    path = T._bpe_path() or "/root/reference/model/bpe_simple_vocab_16e6.txt.gz"
    if not os.path.exists(path):
        pytest.skip("BPE merges table not present on this machine")
    bpe = T.BPETokenizer(path)
    table = json.load(open(os.path.join(GOLDEN, "token_ids.json")))
    for s, ids in table.items():
        assert [T.SOT] + bpe.encode(s) + [T.EOT] == ids, s


def test_class_sentences_and_constants():
    import forward_utils as FU
    normal, abnormal = FU.class_sentences("MVTec", "bottle")
    assert len(normal) == 6 and len(abnormal) == 10
    assert normal[0] == "dark bottle." and abnormal[1] == "a photo of a damaged dark bottle."
    with pytest.raises(AssertionError):
        FU.class_sentences("MVTec", "nope")
    assert FU.DOMAINS["MVTec"] == "Industrial" and FU.DOMAINS["Brain"] == "Medical"


def test_state_dict_contract_tiny():
    from model.model import CLIP
    from model.adapter import AdaptedCLIP
    cfg = synth.tiny_cfg()
    clip = CLIP(cfg.embed_dim,
                dict(image_size=cfg.image_size, layers=cfg.vision.layers, width=cfg.vision.width,
                     patch_size=cfg.patch_size),
                dict(context_length=77, vocab_size=cfg.vocab_size, width=cfg.text.width, heads=cfg.text.heads,
                     layers=cfg.text.layers))
    sd = synth.synth_clip_state_dict(cfg, 7)
    assert set(clip.state_dict().keys()) == set(sd.keys())
    clip.load_state_dict(sd, strict=True)
    m = AdaptedCLIP(clip, relu=False, image_adapt_until=2, text_adapt_until=1, levels=[2, 3])
    assert list(m.image_adapter.state_dict()) == [
        "layer_adapters.0.fc.0.weight", "layer_adapters.1.fc.0.weight", "seg_proj.0.fc.weight",
        "seg_proj.1.fc.weight", "det_proj.fc.weight"]
    assert list(m.text_adapter.state_dict()) == ["0.fc.0.weight", "1.fc.0.weight"]
    m2 = AdaptedCLIP(clip, relu=True, image_adapt_until=1, text_adapt_until=1, levels=[3])
    assert "seg_proj.0.fc.0.weight" in m2.image_adapter.state_dict()
    assert clip.transformer.get_cast_dtype() == torch.float32
    assert clip.visual.grid_size == (5, 5)


def test_no_cpu_fallback():
    from model.model import CLIP
    cfg = synth.tiny_cfg()
    clip = CLIP(cfg.embed_dim,
                dict(image_size=cfg.image_size, layers=cfg.vision.layers, width=cfg.vision.width,
                     patch_size=cfg.patch_size),
                dict(context_length=77, vocab_size=cfg.vocab_size, width=cfg.text.width, heads=cfg.text.heads,
                     layers=cfg.text.layers))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        clip.encode_image(torch.zeros(1, 3, 70, 70), [1])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        clip.encode_text(torch.zeros(1, 77, dtype=torch.int32))


def test_create_model_errors_and_resize(golden_tiny):
    from model.clip import create_model, get_model_config, list_models
    from model.model import resize_pos_embed
    assert "ViT-L-14-336" in list_models()
    cfg = get_model_config("ViT-L-14-336")
    assert cfg["vision_cfg"]["width"] == 1024 and cfg["text_cfg"]["layers"] == 12
    with pytest.raises(RuntimeError, match="not found"):
        create_model("nope", 518)
    with pytest.raises(RuntimeError):
        create_model("ViT-L-14-336", 518, pretrained="openai")  # no checkpoint file in the tree
    import types
    sd = {"visual.positional_embedding": torch.from_numpy(golden_tiny["resize.in"]).clone()}
    fake = types.SimpleNamespace(visual=types.SimpleNamespace(grid_size=(37, 37)))
    resize_pos_embed(sd, fake)
    assert torch.allclose(sd["visual.positional_embedding"], torch.from_numpy(golden_tiny["resize.out"]), atol=1e-6)


def test_precision_mapping(monkeypatch):
    assert engine.dtype_code("fp32") == _lib.F32 and engine.dtype_code("fp16") == _lib.F16
    assert engine.dtype_code("bf16") == _lib.BF16
    monkeypatch.setenv("AACLIP_COMPUTE", "bf16")
    assert engine.dtype_code("fp32") == _lib.BF16
    monkeypatch.delenv("AACLIP_COMPUTE")
    with pytest.raises(ValueError):
        engine.dtype_code("int8")


def test_lds_swizzle_model():
    import subprocess, sys
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "lds_bank_model.py")], check=True)


def test_checkpoint_loading_state_dict_and_torchscript(tmp_path, monkeypatch):
    """reference model/openai.py:56-83 + model/clip.py:62-70: OpenAI files are TorchScript archives with
    fp16 weights and three scalar extras; plain state-dict files load too.  Only tensors are read."""
    from model.clip import create_model, load_checkpoint
    from model.model import CLIP
    from script_archive import save_scripted_state_dict
    cfg = synth.tiny_cfg()
    sd = synth.synth_clip_state_dict(cfg, seed=3)

    def fresh():
        return CLIP(cfg.embed_dim,
                    dict(image_size=cfg.image_size, layers=cfg.vision.layers, width=cfg.vision.width,
                         patch_size=cfg.patch_size),
                    dict(context_length=77, vocab_size=cfg.vocab_size, width=cfg.text.width, heads=cfg.text.heads,
                         layers=cfg.text.layers))
    plain = str(tmp_path / "plain.pt")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, plain)
    m = fresh()
    load_checkpoint(m, plain)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    extras = dict(sd, input_resolution=torch.tensor(70), context_length=torch.tensor(77), vocab_size=torch.tensor(cfg.vocab_size))
    jit = str(tmp_path / "openai.pt")
    save_scripted_state_dict(extras, jit, half=True)
    m = fresh()
    load_checkpoint(m, jit)
    for k, v in m.state_dict().items():
        assert v.dtype == torch.float32 and torch.equal(v, sd[k].half().float()), k
    # create_model(pretrained="openai") picks the file up through AACLIP_CLIP_CKPT; wrong shapes fail strictly
    monkeypatch.setenv("AACLIP_CLIP_CKPT", jit)
    with pytest.raises(RuntimeError):
        create_model("ViT-L-14-336", 518, pretrained="openai")


def test_gemm_k_loop_has_no_waterfall_loops(tmp_path):
    """Regression guard for a silent 15 % loss: when hipcc cannot prove the K loop's buffer descriptors
    wave-uniform it wraps every `buffer_load ... lds` in a waterfall loop (v_readfirstlane x4 + s_and_saveexec).
    The default GEMM kernel's K loop (between its first s_barrier and the last one before the epilogue's first
    store) must contain neither."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(REPO, "aa-clip-iqm_amd", "csrc", "gemm256t.hip")
    out = str(tmp_path / "gemm256t.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    checked = 0
    # NP = 0: plain fp16 operands; NP = 4 / 3: the split-fp16 instantiations (virtual K tiles, run-time tile offsets).
    # For those a spill is worse than slow: a spilled address comes back through scratch_load + s_waitcnt vmcnt(0),
    # which drains the DMA pipeline once per tile (measured: 1.60 -> 1.35 ms per c_fc launch when the last one went).
    # (the last template flag: the QKV product writing the attention kernel's e4m3 records, GemmParams::out_qk8)
    # (the very last flag: the WALKING form -- one workgroup per CU, the next tile's first K tile fetched under the epilogue
    # -- which exists for the split instantiations and is their default.  Its descriptors are loop-carried, which is
    # exactly where hipcc could lose the uniformity proof.  Between the K loop's last barrier and the epilogue's first one it
    # builds the next tile's descriptors with v_readfirstlane on purpose: that stretch is left out.)
    cases = [(e, n, 0, 0) for n in (0, 4, 3) for e in range(5)] + [(0, 4, 1, 0), (0, 3, 1, 0)]
    cases += [(e, n, 0, 1) for n in (4, 3) for e in range(5)] + [(0, 4, 1, 1), (0, 3, 1, 1)]
    for epi, np_, qk8, walk in cases:
        sym = f"_ZN6aaclip18gemm16_256x_kernelIDF16_Li{epi}ELi{np_}ELb{qk8}ELb{walk}EEEvNS_10GemmParamsEiiiii:"
        start = next(i for i, l in enumerate(lines) if l.startswith(sym))
        end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
        body = lines[start:end]
        first_store = next(i for i, l in enumerate(body) if "global_store" in l or "buffer_store" in l)
        bars = [i for i, l in enumerate(body[:first_store]) if "s_barrier" in l]   # (the epilogue has barriers of its own)
        loop = [l.split()[0] for l in body[bars[0]:bars[-2 if walk else -1]] if l.strip() and not l.strip().startswith((".", ";"))]
        assert loop.count("buffer_load_dwordx4") >= 16, (epi, "K loop not found")
        assert "v_readfirstlane_b32" not in loop and "s_and_saveexec_b64" not in loop, \
            f"EPI {epi} NP {np_} walk {walk}: waterfall loop around the K-loop DMA (descriptor not provably uniform)"
        assert not any(op.startswith("scratch_") for op in loop), f"EPI {epi} NP {np_} walk {walk}: spill in the K loop"
        if walk:   # ... and the prefetch of the next tile must not sit in a waterfall loop either
            tail = [l.split()[0] for l in body[bars[-2]:bars[-1]] if l.strip() and not l.strip().startswith((".", ";"))]
            assert tail.count("buffer_load_dwordx4") == 8, (epi, np_, "prefetch of the next tile not found")
            assert "s_and_saveexec_b64" not in tail, f"EPI {epi} NP {np_}: waterfall loop around the next tile's prefetch"
        checked += 1
    assert checked == 29
    # the half-tile kernel (4 waves, 256 x 128): the same two properties for every instantiation the launcher can reach
    for epi, np_, qk8 in [(e, n, 0) for n in (0, 4, 3) for e in range(5)] + [(0, 4, 1), (0, 3, 1)]:
        sym = f"_ZN6aaclip18gemm16_256h_kernelIDF16_Li{epi}ELi{np_}ELb{qk8}EEEvNS_10GemmParamsEiiiii:"
        start = next(i for i, l in enumerate(lines) if l.startswith(sym))
        end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
        body = lines[start:end]
        first_store = next(i for i, l in enumerate(body) if "global_store" in l or "buffer_store" in l)
        bars = [i for i, l in enumerate(body[:first_store]) if "s_barrier" in l]
        loop = [l.split()[0] for l in body[bars[0]:bars[-1]] if l.strip() and not l.strip().startswith((".", ";"))]
        assert loop.count("buffer_load_dwordx4") >= 12, (epi, np_, "K loop not found")
        assert "v_readfirstlane_b32" not in loop and "s_and_saveexec_b64" not in loop, \
            f"half tile, EPI {epi} NP {np_}: waterfall loop around the K-loop DMA"
        assert not any(op.startswith("scratch_") for op in loop), f"half tile, EPI {epi} NP {np_}: spill in the K loop"


def test_tower_run_plans_one_call_with_tap_buffers(monkeypatch):
    """Host logic of Transformer.run (no GPU: engine.run_blocks is replaced by a recorder): the whole tower is ONE
    aaclip_blocks_taps call with one output buffer per block; a tapped buffer is never written again (the block behind
    it continues in a fresh one); adapters are handed to the call block by block; a change of attention mode
    (DAPM_replace) splits the tower into one call per mode, the second reading what the first wrote last."""
    from model.transformer import Transformer
    calls = []

    def fake_run_blocks(x, blocks, B, L, heads, code, causal=False, adapter_weights=None, mix=0.0, x_out=None,
                        x_outs=None):
        assert x_out is None and x_outs is not None and len(x_outs) == len(blocks)
        calls.append({"src": x, "blocks": list(blocks), "outs": list(x_outs),
                      "aws": list(adapter_weights) if adapter_weights is not None else None, "mix": mix})
        for i, t in enumerate(x_outs):          # mark who wrote each buffer last
            t.fill_(float(len(calls) * 100 + i))

    monkeypatch.setattr(engine, "run_blocks", fake_run_blocks)
    tr = Transformer(width=64, layers=6, heads=1, mlp_ratio=2.0)
    x = torch.zeros(4, 64)
    aws = [torch.zeros(1), torch.zeros(1), None, None, None, None]
    final, taps = tr.run(x, 2, 2, _lib.F16, False, [2, 4, 6], adapter_weights=aws, mix=0.1)
    assert len(calls) == 1 and calls[0]["src"] is x and calls[0]["aws"] == aws and calls[0]["mix"] == 0.1
    outs = calls[0]["outs"]
    assert outs[0] is x and outs[1] is x                      # blocks 1, 2 in place on the caller's buffer
    assert outs[2] is outs[3] and outs[2] is not x            # behind the first tap: a fresh buffer
    assert outs[4] is outs[5] and outs[4] is not outs[2] and outs[4] is not x
    assert [t.data_ptr() for t in taps] == [x.data_ptr(), outs[2].data_ptr(), outs[4].data_ptr()]
    assert final is outs[5] and final is taps[2]
    # no taps: everything in place, one call
    calls.clear()
    final, taps = tr.run(torch.zeros(4, 64), 2, 2, _lib.F16, False)
    assert len(calls) == 1 and taps == [] and all(t is calls[0]["src"] for t in calls[0]["outs"]) and final is calls[0]["src"]
    # a mode change after block 4 (stage-1 "surgery" blocks): two calls; the second reads the first one's last buffer
    calls.clear()
    for blk in list(tr.resblocks)[4:]:
        blk.surgery = True
    x = torch.zeros(4, 64)
    final, taps = tr.run(x, 2, 2, _lib.F16, False, [3])
    assert [len(c["blocks"]) for c in calls] == [4, 2]
    assert calls[0]["outs"][2] is x and calls[0]["outs"][3] is not x      # tap after block 3, block 4 in a fresh buffer
    assert calls[1]["src"] is calls[0]["outs"][3] and all(t is calls[1]["src"] for t in calls[1]["outs"])
    assert taps[0] is x and final is calls[1]["outs"][-1]


def test_documented_import_resolution():
    """which of the reference caller's imports (reference test_last.py:13-22) resolve to the
    build.  With the SCRIPT directory first (plain `python /path/to/reference/test_last.py`, PYTHONPATH = build) only
    `model` does -- the reference's model/ is a namespace portion, its dataset/ a regular package, forward_utils.py and
    utils.py plain modules; with the BUILD first (aa-clip-iqm_amd/run_reference_script.py, or the build's own
    test_last.py) all four do."""
    import subprocess
    import sys
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this machine")
    build = os.path.join(REPO, "aa-clip-iqm_amd")
    probe = (
        "import sys, json, importlib.util as u\n"
        "first, second = sys.argv[1], sys.argv[2]\n"
        "sys.path[:] = [first, second] + [p for p in sys.path if p not in ('', first, second)]\n"
        "out = {}\n"
        "for name in ('model', 'dataset', 'forward_utils', 'utils'):\n"
        "    s = u.find_spec(name)\n"
        "    loc = s.origin if s.origin else list(s.submodule_search_locations)[0]\n"
        "    out[name] = loc\n"
        "print(json.dumps(out))\n")

    def resolve(first, second):
        r = subprocess.run([sys.executable, "-c", probe, first, second], capture_output=True, text=True, check=True,
                           env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
        return {k: ("build" if v.startswith(build) else "reference" if v.startswith(ref) else v)
                for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}

    assert resolve(ref, build) == {"model": "build", "dataset": "reference", "forward_utils": "reference",
                                   "utils": "reference"}
    assert resolve(build, ref) == {"model": "build", "dataset": "build", "forward_utils": "build", "utils": "build"}
    # the launcher really orders sys.path that way (no GPU needed: the probe script only prints the resolution)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for name in ("dataset", "model"):
            os.makedirs(os.path.join(d, name))
        open(os.path.join(d, "dataset", "__init__.py"), "w").write("WHO = 'caller'\n")
        open(os.path.join(d, "forward_utils.py"), "w").write("WHO = 'caller'\n")
        open(os.path.join(d, "utils.py"), "w").write("WHO = 'caller'\n")
        script = os.path.join(d, "caller.py")
        open(script, "w").write(
            "import sys, importlib.util as u\n"
            "print('RESOLVED', [u.find_spec(n).origin for n in ('dataset', 'forward_utils', 'utils')], sys.argv[1:])\n")
        r = subprocess.run([sys.executable, os.path.join(build, "run_reference_script.py"), script, "--flag", "7"],
                           capture_output=True, text=True, check=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESOLVED")][-1]
        assert line.count(build) == 3 and d not in line.split("]")[0] and "['--flag', '7']" in line


def test_reference_callers_names_exist_in_the_build():
    """Every name reference test_last.py:13-22 imports exists in the build's modules (visualize as a stub that says it
    is out of scope when CALLED)."""
    import forward_utils as FU
    import utils as U
    from dataset import DOMAINS, get_dataset  # noqa: F401
    from model.adapter import AdaptedCLIP  # noqa: F401
    from model.clip import create_model  # noqa: F401
    for n in ("get_adapted_text_embedding", "calculate_similarity_map", "metrics_eval", "visualize"):
        assert callable(getattr(FU, n))
    with pytest.raises(NotImplementedError):
        FU.visualize()
    U.setup_seed(3)
    a = torch.nn.functional.normalize(torch.randn(5, 8), dim=-1)
    assert torch.allclose(U.cos_sim(a[0], a), a @ a[0]) and U.cos_sim(a, a).shape == (5, 5)


def test_split_row_formats_on_the_host():
    """engine.split_rows / split16_rows (the AACLIP_F16X2 operand formats, include/aaclip.h): plane layout, scales and
    what a product sees of a value; weights exact in fp16 take the 3-plane form only when asked and K % 256 == 0."""
    x = synth.randn("t.split.host", (5, 256), 2.0, 1)
    s8 = engine.split_rows(x)
    assert s8.dtype == torch.uint8 and s8.shape == (5, 1024)
    hi = s8[:, :512].contiguous().view(torch.float16)
    assert torch.equal(hi, x.half())
    lo8 = s8[:, 512:768].contiguous().view(torch.float8_e4m3fn).float() / 1024.0
    hi8 = s8[:, 768:].contiguous().view(torch.float8_e4m3fn).float()
    assert float(((hi.float() + lo8 - x).abs() / (x.abs() * 2.0 ** -15 + 2.0 ** -19)).max()) <= 1.0
    assert float(((hi8 - x).abs() / (x.abs() * 2.0 ** -4 + 2.0 ** -9)).max()) <= 1.0
    assert torch.allclose(engine.join_split8(s8, 256), (hi.float() + lo8).double())
    s16 = engine.split16_rows(x)
    assert s16.dtype == torch.float16 and s16.shape == (5, 512)
    assert float((s16[:, :256].double() + s16[:, 256:].double() - x.double()).abs().max()) <= 2.0 ** -20
    w = engine.split_rows(x, weight=True)
    wh8 = w[:, 512:768].contiguous().view(torch.float8_e4m3fn).float() / 64.0
    assert float(((wh8 - x).abs() / (x.abs() * 2.0 ** -4 + 2.0 ** -15)).max()) <= 1.0
    # saturation instead of NaN beyond e4m3's range (the hardware conversion would return NaN: the kernels clamp too)
    big = engine.split_rows(torch.tensor([[1000.0, -1e6] + [0.0] * 254]))
    assert float(big[:, 768:].contiguous().view(torch.float8_e4m3fn).float().abs().max()) == 448.0
    p = torch.nn.Parameter(x.half().float().repeat(1, 1))
    assert engine.CACHE.get(p, _lib.F16X2, "plain+exact").shape == (5, 768)        # exact in fp16, K = 256
    assert engine.CACHE.get(p, _lib.F16X2, "plain").shape == (5, 1024)
    q = torch.nn.Parameter(x[:, :128].half().float().contiguous())
    assert engine.CACHE.get(q, _lib.F16X2, "plain+exact").shape == (5, 512)        # K = 128: pairs of K tiles, no 3-plane form
    assert engine.dtype_code("fp16x2") == _lib.F16X2 and engine.plain_code(_lib.F16X2) == _lib.F32


def test_iqm_cross_attention_foldings_agree_on_cpu(monkeypatch):
    """Host algebra of the IQM visual cross-attention (model/iqm.py IQM._attend), with every engine call replaced by its
    torch definition on the CPU: (a) the reference's order -- project every row through query_adapters, torch.cat,
    visual_feature_proj, W_k and W_v, then attend per head (reference model/adapter.py:205-221, model/iqm.py:108-139);
    (b) W_k, W_v and visual_feature_proj folded onto the effective queries (aaclip_cross_rows on the concatenated rows);
    (c) query_adapters folded as well (aaclip_cross_rows_levels on the LayerNorm'ed tap rows, first row of every image
    skipped).  All three must give the same attention output: the weights w_in / w_out that AdaptedCLIP._iqm_levels
    concatenates, the [B, R, level, D] layout and the row0 / rows_per_image bookkeeping are what this pins."""
    import math
    from model.iqm import IQM
    F32 = _lib.F32

    def fake_gemm(code, epi, a, w, bias, out, act=0):
        y = a.double() @ w.double().t()
        if bias is not None:
            y = y + bias.double()
        if epi == _lib.EPI_BIAS_GELU:
            y = 0.5 * y * (1 + torch.erf(y / math.sqrt(2)))
        out.copy_(y.to(out.dtype))
        return out

    class FakeCache:
        def get(self, w, code, kind=None):
            w = w.detach().float()
            return w.t().contiguous() if kind == "transpose" else w

    def fake_head_expand(q, H, scale, code):
        rows, D = q.shape
        hd = D // H
        out = torch.zeros(rows, H, D)
        for h in range(H):
            out[:, h, h * hd:(h + 1) * hd] = q[:, h * hd:(h + 1) * hd] * scale
        return out.view(rows * H, D)

    def fake_head_diag(full, H):
        rows, D = full.shape[0] // H, full.shape[1]
        hd = D // H
        f = full.view(rows, H, D)
        return torch.cat([f[:, h, h * hd:(h + 1) * hd] for h in range(H)], 1).contiguous()

    def fake_cross_rows(qt, x, B, R, Lk, code):
        Dk = x.shape[-1]
        p = torch.softmax(qt.double().view(B, R, Dk) @ x.double().view(B, Lk, Dk).transpose(1, 2), -1)
        return (p @ x.double().view(B, Lk, Dk)).float().view(B * R, Dk)

    def fake_cross_rows_levels(qt, levels, B, R, rpi, row0, Lk, Dk):
        n = len(levels)
        q = qt.double().view(B, R, n, Dk)
        keys = [x.double().view(B, rpi, -1)[:, row0:row0 + Lk, :Dk] for x in levels]
        p = torch.softmax(torch.cat([torch.einsum("brd,bjd->brj", q[:, :, s], keys[s]) for s in range(n)], -1), -1)
        out = torch.stack([torch.einsum("brj,bjd->brd", p[:, :, s * Lk:(s + 1) * Lk], keys[s]) for s in range(n)], 2)
        return out.float().reshape(B * R, n * Dk)

    def fake_small_attention(q, k, v, B, nq, Lk, H, code):
        D = q.shape[-1]
        hd = D // H
        qh = q.double().view(B, nq, H, hd).transpose(1, 2)
        kh = k.double().view(B, Lk, H, hd).transpose(1, 2)
        vh = v.double().view(B, Lk, H, hd).transpose(1, 2)
        return (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), -1) @ vh).transpose(1, 2).reshape(B * nq, D).float()

    def fake_res_ln(a, b, ln, eps):
        x = a if b is None else a + b
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, eps)

    monkeypatch.setattr(engine, "gemm", fake_gemm)
    monkeypatch.setattr(engine, "CACHE", FakeCache())
    monkeypatch.setattr(engine, "head_expand", fake_head_expand)
    monkeypatch.setattr(engine, "head_diag", fake_head_diag)
    monkeypatch.setattr(engine, "cross_rows", fake_cross_rows)
    monkeypatch.setattr(engine, "cross_rows_levels", fake_cross_rows_levels)
    monkeypatch.setattr(engine, "small_attention", fake_small_attention)
    monkeypatch.setattr(engine, "residual_layernorm", fake_res_ln)

    torch.manual_seed(3)
    B, nq, H, hid, Dtap, L, nlev = 2, 2, 4, 256, 48, 9, 3          # 8 effective queries, 3 levels of 8 patch rows
    iqm = IQM(hidden_size=hid, num_hidden_layers=1, num_attention_heads=H, encoder_hidden_size=hid,
              text_encoder_hidden_size=hid, intermediate_size=64)
    att = iqm.encoder.layer[0].crossattention
    with torch.no_grad():
        for prm in att.parameters():
            prm.normal_(0, 0.2 if prm.dim() > 1 else 0.3)
        att.output.LayerNorm.weight.add_(1.0)
    wqa = [torch.randn(hid, Dtap) * 0.3 for _ in range(nlev)]                 # query_adapters[k].weight (no bias)
    P, bp = torch.randn(hid, hid) * 0.1, torch.randn(hid) * 0.2              # visual_feature_proj
    taps = [torch.randn(B * L, Dtap) for _ in range(nlev)]                    # ln_post(tap k), CLS row included
    h = torch.randn(B * nq, hid)
    with torch.no_grad():
        # (a) the reference's order: rows projected level by level, concatenated per image, projected again
        vis = torch.cat([(t.view(B, L, Dtap)[:, 1:] @ w.t()) for t, w in zip(taps, wqa)], 1)          # [B, nlev*(L-1), hid]
        enc = (vis @ P.t() + bp).reshape(-1, hid)
        Lk = nlev * (L - 1)
        # (_attend itself takes the folded branch whenever rows of this width are given, so the unfolded order is written out)
        q = h @ att.attention.query.weight.t() + att.attention.query.bias
        k = enc @ att.attention.key.weight.t() + att.attention.key.bias
        v = enc @ att.attention.value.weight.t() + att.attention.value.bias
        ctx = fake_small_attention(q, k, v, B, nq, Lk, H, F32)
        a = fake_res_ln(ctx @ att.output.dense.weight.t() + att.output.dense.bias, h, att.output.LayerNorm, iqm.eps)
        # (b) W_k, W_v, visual_feature_proj folded; rows = the concatenated query_adapters outputs
        b = iqm._attend(att, h, vis.reshape(-1, hid).contiguous(), B, nq, Lk, F32, enc_proj=(P, bp))
        # (c) query_adapters folded too: rows = the LayerNorm'ed taps themselves
        lv = {"rows": taps, "rows_per_image": L, "row0": 1, "keys": L - 1, "width": Dtap,
              "w_in": torch.cat([w.t() for w in wqa], 0).contiguous(), "w_out": torch.cat(wqa, 1).contiguous()}
        c = iqm._attend(att, h, None, B, nq, Lk, F32, enc_proj=(P, bp), enc_levels=lv)
    assert a.shape == b.shape == c.shape == (B * nq, hid)
    assert float(a.abs().max()) > 0.5
    assert torch.allclose(b, a, atol=2e-5, rtol=1e-5), float((b - a).abs().max())
    assert torch.allclose(c, a, atol=2e-5, rtol=1e-5), float((c - a).abs().max())

    # (d) the anchor rows: 16-bit rows of width 768 read as they are (one segment of the matrix-core row kernel)
    torch.manual_seed(4)
    iq2 = IQM(hidden_size=128, num_hidden_layers=1, num_attention_heads=4, encoder_hidden_size=128,
              text_encoder_hidden_size=768, intermediate_size=64)
    at2 = iq2.encoder.layer[0].text_crossattention
    with torch.no_grad():
        for prm in at2.parameters():
            prm.normal_(0, 0.1 if prm.dim() > 1 else 0.3)
        at2.output.LayerNorm.weight.add_(1.0)
    Lt = 11
    rows = (torch.randn(B * Lt, 768) * 0.7).half()
    h2 = torch.randn(B * nq, 128)
    with torch.no_grad():
        q = h2 @ at2.attention.query.weight.t() + at2.attention.query.bias
        k = rows.float() @ at2.attention.key.weight.t() + at2.attention.key.bias
        v = rows.float() @ at2.attention.value.weight.t() + at2.attention.value.bias
        ctx = fake_small_attention(q, k, v, B, nq, Lt, 4, F32)
        want = fake_res_ln(ctx @ at2.output.dense.weight.t() + at2.output.dense.bias, h2, at2.output.LayerNorm, iq2.eps)
        got = iq2._attend(at2, h2, rows, B, nq, Lt, F32)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())
