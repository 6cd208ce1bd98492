"""ctypes binding of libaaclip_hip.so (C ABI: include/aaclip.h).

The product path has no fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  Build it with `__graft_entry__.build()` or
`make -C aa-clip-iqm_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AACLIP_LIB") or os.path.join(_HERE, "libaaclip_hip.so")   # AACLIP_LIB: experiment builds
MEASURE_LIB_PATH = os.path.join(_HERE, "libaaclip_hip_measure.so")   # `make measure`: A/B variants, ablations, stamps
ABI_VERSION = 5   # include/aaclip.h AACLIP_ABI_VERSION this binding was written against

F32, F16, BF16, F16X2 = 0, 1, 2, 3   # F16X2: split fp16 (hi + lo pairs), include/aaclip.h
EXACT16_QKV, EXACT16_OUT, EXACT16_FC, EXACT16_PROJ, EXACT16_ADAPTER = 1, 2, 4, 8, 16
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_ACT_F32 = 0, 1, 2, 3

_vp, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class BlockWeights(C.Structure):
    """struct aaclip_block_weights (include/aaclip.h).  struct_bytes is filled in on construction; the library
    refuses a struct whose size field does not match its own sizeof."""
    _fields_ = [("struct_bytes", _sz)] + [(n, _vp) for n in (
        "ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b",
        "fc_w", "fc_b", "proj_w", "proj_b", "adapter_w", "fc_w_fold", "fc_fold_s", "fc_fold_b",
        "qkv_w_fold", "qkv_fold_s", "qkv_fold_b")] + [("exact16", C.c_uint)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_bytes = C.sizeof(BlockWeights)


# name -> (restype, argtypes); every symbol include/aaclip.h declares
SIGNATURES = {
    "aaclip_version": (_i, []),
    "aaclip_last_error": (C.c_char_p, []),
    "aaclip_is_measurement_build": (_i, []),
    "aaclip_workspace_bytes": (_sz, [_i, _l, _i, _i, _i]),
    "aaclip_patch_embed": (_i, [_vp] * 7 + [_i] * 6 + [_vp, _sz, _vp]),
    "aaclip_block": (_i, [_vp, C.POINTER(BlockWeights), _f] + [_i] * 7 + [_vp, _sz, _vp]),
    "aaclip_blocks": (_i, [_vp, C.POINTER(BlockWeights), _i, _f] + [_i] * 7 + [_vp, _sz, _vp]),
    "aaclip_blocks_to": (_i, [_vp, _vp, C.POINTER(BlockWeights), _i, _f] + [_i] * 7 + [_vp, _sz, _vp]),
    "aaclip_blocks_taps": (_i, [_vp, C.POINTER(_vp), C.POINTER(BlockWeights), _i, _f] + [_i] * 7 + [_vp, _sz, _vp]),
    "aaclip_tap_head": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp] + [_i] * 5 + [_vp, _sz, _vp]),
    "aaclip_tap_head_keep_rows": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp] + [_i] * 5 + [_vp, _sz, _vp]),
    "aaclip_det_head": (_i, [_vp, _vp, _vp, _vp, _i, _vp] + [_i] * 5 + [_vp, _sz, _vp]),
    "aaclip_anomaly_map": (_i, [C.POINTER(_vp), _i, _vp, _l, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "aaclip_similarity_map_train": (_i, [_vp, _vp, _l, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "aaclip_text_embed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "aaclip_resample_ksize": (_i, [_i, _i]),
    "aaclip_resample_table": (_i, [_i, _i, _vp, _vp]),
    "aaclip_preprocess": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aaclip_row_head": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp] + [_i] * 6 + [_vp, _sz, _vp]),
    "aaclip_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _l, _i, _f, _vp]),
    "aaclip_gemm": (_i, [_i, _i, _vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _f, _vp]),
    "aaclip_attention": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "aaclip_attention_log2q": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "aaclip_adapter_mix": (_i, [_vp, _vp, _l, _i, _f, _vp]),
    "aaclip_small_attention": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "aaclip_cross_rows_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "aaclip_cross_rows": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "aaclip_cross_rows_levels_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "aaclip_cross_rows_levels": (_i, [_i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _l, _vp, _sz, _vp]),
    "aaclip_head_expand": (_i, [_i, _vp, _vp, _l, _i, _i, _f, _vp]),
    "aaclip_head_diag": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "aaclip_residual_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "aaclip_combine3": (_i, [_vp, _vp, _vp, _f, _f, _f, _vp, _l, _vp]),
    "aaclip_linear_smallk": (_i, [_i, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "aaclip_drop_cls_rows": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "aaclip_iqm_map": (_i, [C.POINTER(_vp), _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _sz, _vp]),
    "aaclip_set_gemm_variant": (_i, [_i]),
    "aaclip_debug_gemm_stamps": (_i, [C.POINTER(C.c_double), _i]),
    "aaclip_profile_begin": (_i, [C.c_uint, _i]),
    "aaclip_profile_end": (_i, [C.POINTER(C.c_float), C.POINTER(C.c_int), _i]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the AA-CLIP HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C aa-clip-iqm_amd/csrc`). "
                "There is no PyTorch fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.aaclip_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                               "(make -C aa-clip-iqm_amd/csrc)")
        _lib = lib
    return _lib


def kernel_source_revision(files=("gemm256t.hip", "gemm.hip", "common.h", "mma16.h", "kernels.h")) -> str:
    """sha256 (first 16 hex digits) over the sources of the large-batch GEMM kernels, in the order given.  Offline
    counter measurements committed under profiles/ record it, and bench.py only reports a measurement taken on the
    sources it is running (a stale file yields traffic: null)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(os.path.dirname(_HERE), "csrc")
    for name in files:
        with open(os.path.join(src, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().aaclip_last_error()
        raise RuntimeError(f"aaclip {what} failed (rc={rc}): {msg.decode() if msg else ''}")
