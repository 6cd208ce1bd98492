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


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    assert lib.aaclip_version() == 2
    header = open(os.path.join(REPO, "include", "aaclip.h")).read()
    declared = set(re.findall(r"\b(aaclip_[a-z_0-9]+)\s*\(", header))
    declared.discard("aaclip_block_weights")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


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
    The default GEMM kernel's K loop (between its first and last s_barrier) must contain neither."""
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
    for epi in range(5):
        sym = f"_ZN6aaclip18gemm16_256x_kernelIDF16_Li{epi}EEEvNS_10GemmParamsEiiiii:"
        start = next(i for i, l in enumerate(lines) if l.startswith(sym))
        end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
        body = lines[start:end]
        bars = [i for i, l in enumerate(body) if "s_barrier" in l]
        loop = [l.split()[0] for l in body[bars[0]:bars[-1]] if l.strip() and not l.strip().startswith((".", ";"))]
        assert loop.count("buffer_load_dwordx4") >= 16, (epi, "K loop not found")
        assert "v_readfirstlane_b32" not in loop and "s_and_saveexec_b64" not in loop, \
            f"EPI {epi}: waterfall loop around the K-loop DMA (descriptor not provably uniform)"
        assert not any(op.startswith("scratch_") for op in loop), f"EPI {epi}: spill in the K loop"
        checked += 1
    assert checked == 5
