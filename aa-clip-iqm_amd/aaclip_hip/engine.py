"""Host-side executor: turns torch tensors into raw HIP pointers and sequences
the C-ABI calls of libaaclip_hip.so.  PyTorch is used for device memory, the
current stream and parameter containers; every per-token computation is a kernel
of the library.  The torch ops that do compute are load-time weight preparation
(WeightCache: dtype conversion / transposes; FoldCache: W*gamma, its row sums and
b + W@beta, once per parameter version) -- and, outside this module, the <=16-row
anchor mean and the [B,768].[768] image score in forward_utils.py.

Layout: the residual stream is batch-first, fp32, [B*L, D] contiguous for the
whole tower (the reference permutes to LND, model/adapter.py:158; token rows are
independent so the layout is free).  Matrix-product weights are converted once to
the compute dtype and cached per parameter version.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import BF16, F16, F16X2, F32, BlockWeights

_TORCH_DT = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16, F16X2: torch.float16}
# "fp16x2": split fp16 (every matrix-product operand as an fp16 hi + lo pair, 3 MFMA products per matrix product):
# the 16-bit-MFMA mode whose taps and anomaly maps stay inside 1e-3 abs + 1e-2 rel of the fp32 reference
_PRECISION = {"fp32": F32, "f32": F32, "fp16": F16, "f16": F16, "bf16": BF16, "amp": F16, "pure_fp16": F16,
              "pure_bf16": BF16, "amp_bf16": BF16, "fp16x2": F16X2, "f16x2": F16X2, "split": F16X2}


def dtype_code(precision) -> int:
    """Map a create_model(precision=...) string (reference model/clip.py:88,
    model/model.py:63-69) to the arithmetic type of the matrix products.  The env
    var AACLIP_COMPUTE overrides it for callers that cannot pass precision."""
    env = os.environ.get("AACLIP_COMPUTE")
    if env:
        precision = env
    if isinstance(precision, int):
        return precision
    try:
        return _PRECISION[str(precision).lower()]
    except KeyError:
        raise ValueError(f"unknown precision {precision!r}; use fp32, fp16x2, fp16 or bf16")


def torch_dtype(code: int) -> torch.dtype:
    return _TORCH_DT[code]


def plain_code(code: int) -> int:
    """The arithmetic type of the side paths that have no split-fp16 GEMM kernels (IQM branch): fp32 products there.
    (Not everything on that branch is then exact fp32: with the tap levels folded, the visual cross-attention reads the
    fp16 halves of the split8 tap rows and runs p.v with fp16 probabilities, csrc/iqm.hip cross_rows_mfma_kernel.)"""
    return F32 if code == F16X2 else code


CROSS_ROWS_MAX_SEGMENTS = 4      # csrc/iqm.hip cross_rows_levels_check: tap levels one aaclip_cross_rows_levels call takes


# split fp16 (include/aaclip.h AACLIP_F16X2, csrc/common.h): fixed power-of-two scales of the e4m3 correction planes
SPLIT8_ACT_LO_EXP, SPLIT8_ACT_HI_EXP, SPLIT8_W_HI_EXP, SPLIT8_W_LO_EXP = 10, 0, 6, 17


def _e4m3_bytes(v: torch.Tensor, exp: int) -> torch.Tensor:
    """fp32 -> e4m3 (OCP e4m3fn) bytes of v * 2^exp, clamped to +-448, on v's own device (torch's conversion gives the
    same bytes on the host and on the MI355X; load-time work)."""
    x = (v.detach().float() * float(2 ** exp)).clamp_(-448.0, 448.0)
    return x.to(torch.float8_e4m3fn).view(torch.uint8)


def split16_rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 [R, C] -> split16 rows [R, 2C] fp16: hi = fp16(v) | lo = fp16(v - hi) -- the attention kernel's q|k|v
    input format (what the QKV epilogue writes)."""
    v = t.detach().float()
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    return torch.cat([hi, lo], dim=-1).contiguous()


def split_rows(t: torch.Tensor, weight: bool = False, exact: bool = False) -> torch.Tensor:
    """fp32 [R, C] -> split8 rows as uint8 [R, 4C]: [hi: C x fp16][lo8: C x e4m3][hi8: C x e4m3], the GEMM operand format
    of AACLIP_F16X2 (lo8 = e4m3((v - hi) * 2^10), hi8 = e4m3(v) for activations; weights: [Wh][e4m3(W * 2^6)]
    [e4m3((W - Wh) * 2^17)], the last plane dropped when `exact`).  Load-time / caller-side operand preparation, like
    the .to(dtype) of the other modes; the hot path's own split rows are written by the kernels' epilogues."""
    v = t.detach().float()
    hi = v.to(torch.float16)
    lo = v - hi.float()
    planes = [hi.contiguous().view(torch.uint8).reshape(v.shape[0], -1)]
    if weight:
        planes.append(_e4m3_bytes(v, SPLIT8_W_HI_EXP))
        if not exact:
            planes.append(_e4m3_bytes(lo, SPLIT8_W_LO_EXP))
    else:
        planes += [_e4m3_bytes(lo, SPLIT8_ACT_LO_EXP), _e4m3_bytes(v, SPLIT8_ACT_HI_EXP)]
    return torch.cat(planes, dim=1).contiguous()


def join_split8(t: torch.Tensor, C: int) -> torch.Tensor:
    """split8 rows (uint8 [R, 4C] or fp16 [R, 2C]) -> fp64 hi + lo8 * 2^-10 (what a product sees of an activation)."""
    b = t.contiguous().view(torch.uint8).reshape(t.shape[0], -1).cpu()
    hi = b[:, : 2 * C].contiguous().view(torch.float16).double()
    lo = b[:, 2 * C: 3 * C].contiguous().view(torch.float8_e4m3fn).double() / float(2 ** SPLIT8_ACT_LO_EXP)
    return hi + lo


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on {t.device}; the AA-CLIP HIP path only runs on an MI355X (cuda) device "
            "and has no CPU fallback")


class Workspace:
    """One growing scratch buffer per device (caller-owned as far as the C ABI
    is concerned)."""
    _bufs: Dict[tuple, torch.Tensor] = {}

    @classmethod
    def get(cls, dev: torch.device, nbytes: int) -> torch.Tensor:
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        idx = (idx, torch.cuda.current_stream(dev).cuda_stream)   # one scratch buffer per (device, stream)
        buf = cls._bufs.get(idx)
        if buf is None or buf.numel() < nbytes:
            cls._bufs[idx] = None
            buf = torch.empty(int(nbytes * 1.02) + 4096, dtype=torch.uint8, device=dev)
            cls._bufs[idx] = buf
        return buf

    @classmethod
    def for_rows(cls, dev: torch.device, code: int, rows: int, D: int, F: int, E: int) -> torch.Tensor:
        n = _lib.load().aaclip_workspace_bytes(code, rows, D, F, E)
        return cls.get(dev, n)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class WeightCache:
    """Converted copies of matrix weights, one per (parameter object, dtype, kind),
    refreshed when the parameter's storage or version changes (adapters are
    trainable).  Entries hold a weak reference to the parameter: id() values and
    device addresses are recycled once a model is freed, so identity is checked
    on the object itself and dead entries are dropped."""

    def __init__(self):
        self._c: Dict[Tuple[int, int, str], tuple] = {}

    def get(self, p: torch.Tensor, code: int, kind: str = "plain") -> torch.Tensor:
        """kind: 'plain' | 'transpose' | 'conv'.  Code F16X2: split8 weight rows as uint8 [out, 4*in]; with a '+exact'
        suffix the 3-plane form [out, 3*in] when every value is exact in fp16 and in_features is a multiple of 256
        (the kernels then skip the weight-lo correction tile) -- tell them apart by the shape."""
        key = (id(p), code, kind)
        hit = self._c.get(key)
        if hit is not None and hit[0]() is p and hit[1] == p.data_ptr() and hit[2] == p._version:
            return hit[3]
        src = p.detach()
        allow_exact = kind.endswith("+exact")
        kind = kind.split("+")[0]
        if kind == "transpose":          # [in, out] parameter used as x @ P  ->  [out, in]
            src = src.t()
        elif kind == "conv":             # conv1.weight [D,3,ps,ps] -> [D, Kpad]
            d = src.shape[0]
            flat = src.reshape(d, -1)
            kpad = (flat.shape[1] + 63) // 64 * 64
            pad = torch.zeros(d, kpad, dtype=flat.dtype, device=flat.device)
            pad[:, : flat.shape[1]] = flat
            src = pad
        if code == F16X2:
            src32 = src.float()
            exact = (allow_exact and src.shape[1] % 256 == 0
                     and bool((src32.to(torch.float16).float() == src32).all()))
            out = split_rows(src32, weight=True, exact=exact)
        else:
            out = src.to(_TORCH_DT[code]).contiguous()
        cache = self._c

        def _drop(_ref, key=key):
            ent = cache.get(key)
            if ent is not None and ent[0] is _ref:
                del cache[key]

        self._c[key] = (weakref.ref(p, _drop), p.data_ptr(), p._version, out)
        return out


CACHE = WeightCache()


class FoldCache:
    """ln_2 folded into c_fc (include/aaclip.h, aaclip_block_weights): per block and dtype
    (c_fc.weight * ln_2.weight in the compute dtype, its row sums, c_fc.bias + c_fc.weight @ ln_2.bias),
    rebuilt when any of the four parameters changes."""

    def __init__(self):
        self._c: Dict[Tuple[int, int], tuple] = {}

    @staticmethod
    def _fold(weight, bias, gamma, beta, code):
        fw, fb, g, be = (p.detach().float() for p in (weight, bias, gamma, beta))
        wf = (fw * g[None, :]).to(_TORCH_DT[code]).contiguous()
        return wf, wf.float().sum(dim=1).contiguous(), (fb + fw @ be).contiguous()

    def get(self, block, code: int):
        """-> (fc_w_fold, fc_fold_s, fc_fold_b, qkv_w_fold, qkv_fold_s, qkv_fold_b)"""
        ps = (block.mlp.c_fc.weight, block.mlp.c_fc.bias, block.ln_2.weight, block.ln_2.bias,
              block.attn.in_proj_weight, block.attn.in_proj_bias, block.ln_1.weight, block.ln_1.bias)
        sig = tuple((p.data_ptr(), p._version) for p in ps)
        key = (id(block), code)
        hit = self._c.get(key)
        if hit is not None and hit[0]() is block and hit[1] == sig:
            return hit[2]
        with torch.no_grad():
            out = self._fold(*ps[:4], code) + self._fold(*ps[4:], code)
        cache = self._c

        def _drop(_ref, key=key):
            ent = cache.get(key)
            if ent is not None and ent[0] is _ref:
                del cache[key]

        self._c[key] = (weakref.ref(block, _drop), sig, out)
        return out


FOLDS = FoldCache()


def _keep(refs: list, t: torch.Tensor) -> int:
    refs.append(t)
    return t.data_ptr()


def pack_block(block, code: int, adapter_weight: Optional[torch.Tensor]) -> Tuple[BlockWeights, list]:
    """Build the aaclip_block_weights struct for one ResidualAttentionBlock module."""
    refs: list = []
    w = BlockWeights()
    w.ln1_w = _keep(refs, _f32c(block.ln_1.weight))
    w.ln1_b = _keep(refs, _f32c(block.ln_1.bias))
    ex = 0

    def mat(param, bit):
        """matrix weight in the compute dtype; split fp16: plain fp16 + its exact16 bit when the lo half is zero"""
        nonlocal ex
        if code != F16X2:
            return _keep(refs, CACHE.get(param, code))
        t = CACHE.get(param, code, "plain+exact")
        if t.shape[1] == 3 * param.shape[1]:
            ex |= bit
        return _keep(refs, t)

    w.qkv_w = mat(block.attn.in_proj_weight, _lib.EXACT16_QKV)
    w.qkv_b = _keep(refs, _f32c(block.attn.in_proj_bias))
    w.out_w = mat(block.attn.out_proj.weight, _lib.EXACT16_OUT)
    w.out_b = _keep(refs, _f32c(block.attn.out_proj.bias))
    w.ln2_w = _keep(refs, _f32c(block.ln_2.weight))
    w.ln2_b = _keep(refs, _f32c(block.ln_2.bias))
    w.fc_w = mat(block.mlp.c_fc.weight, _lib.EXACT16_FC)
    w.fc_b = _keep(refs, _f32c(block.mlp.c_fc.bias))
    w.proj_w = mat(block.mlp.c_proj.weight, _lib.EXACT16_PROJ)
    w.proj_b = _keep(refs, _f32c(block.mlp.c_proj.bias))
    w.adapter_w = mat(adapter_weight, _lib.EXACT16_ADAPTER) if adapter_weight is not None else None
    w.exact16 = ex
    if code in (F16, BF16):
        wf, fs, fb, qf, qs, qb = FOLDS.get(block, code)
        w.fc_w_fold, w.fc_fold_s, w.fc_fold_b = _keep(refs, wf), _keep(refs, fs), _keep(refs, fb)
        w.qkv_w_fold, w.qkv_fold_s, w.qkv_fold_b = _keep(refs, qf), _keep(refs, qs), _keep(refs, qb)
    return w, refs


# ----------------------------------------------------------------------------
# path-level calls
# ----------------------------------------------------------------------------
def patch_embed(img: torch.Tensor, visual, code: int) -> Tuple[torch.Tensor, int, int]:
    """reference model/adapter.py:139-156 -> x [B*L, D] fp32, returns (x, B, L)."""
    require_gpu(img, "patch_embed")
    lib = _lib.load()
    img = _f32c(img)
    B, Cc, H, W = img.shape
    if Cc != 3:
        raise ValueError("patch_embed expects [B,3,H,W] images")
    ps = visual.patch_size[0]
    D = visual.embed_dim
    L = (H // ps) * (W // ps) + 1
    pos = _f32c(visual.positional_embedding)
    if pos.shape[0] != L:
        raise RuntimeError(f"positional_embedding has {pos.shape[0]} rows but the image needs {L}")
    x = torch.empty(B * L, D, dtype=torch.float32, device=img.device)
    ws = Workspace.for_rows(img.device, code, B * L, D, 4 * D, 0)
    conv_w = CACHE.get(visual.conv1.weight, code, "conv")
    cls, lw, lb = _f32c(visual.class_embedding), _f32c(visual.ln_pre.weight), _f32c(visual.ln_pre.bias)
    _lib.check(lib.aaclip_patch_embed(img.data_ptr(), conv_w.data_ptr(), cls.data_ptr(), pos.data_ptr(),
                                      lw.data_ptr(), lb.data_ptr(), x.data_ptr(), B, H, W, ps, D, code,
                                      ws.data_ptr(), ws.numel(), _stream(img.device)), "patch_embed")
    return x, B, L


ATTN_FULL, ATTN_CAUSAL, ATTN_VV_BATCH = 0, 1, 2


def run_block(x: torch.Tensor, block, B: int, L: int, heads: int, code: int, causal: bool = False,
              adapter_weight: Optional[torch.Tensor] = None, mix: float = 0.0) -> None:
    """In place on x [B*L, D]: reference model/transformer.py:239-258 (+ adapter.py:163-170).
    A block flagged by VisionTransformer.DAPM_replace (`block.surgery`) runs the V-V attention over
    the batch axis (reference transformer.py:102-152 as executed, include/aaclip.h AACLIP_ATTN_VV_BATCH)."""
    if getattr(block, "surgery", False):
        if causal:
            raise ValueError("the V-V attention block takes no mask")
        causal = ATTN_VV_BATCH
    require_gpu(x, "block")
    lib = _lib.load()
    D = x.shape[1]
    F = block.mlp.c_fc.weight.shape[0]
    w, refs = pack_block(block, code, adapter_weight)
    ws = Workspace.for_rows(x.device, code, B * L, D, F, 0)
    _lib.check(lib.aaclip_block(x.data_ptr(), C.byref(w), float(mix), B, L, D, heads, F, int(causal), code,
                                ws.data_ptr(), ws.numel(), _stream(x.device)), "block")
    del refs


def run_blocks(x: torch.Tensor, blocks: Sequence, B: int, L: int, heads: int, code: int, causal: bool = False,
               adapter_weights: Optional[Sequence[Optional[torch.Tensor]]] = None, mix: float = 0.0,
               x_out: Optional[torch.Tensor] = None, x_outs: Optional[Sequence[torch.Tensor]] = None) -> None:
    """Consecutive blocks in ONE aaclip_blocks call (in place on x [B*L, D]); nothing reads x in between, so
    the library folds ln_1 of every block but the first into its QKV product.  Blocks flagged by
    DAPM_replace run their V-V attention; a run must not mix the two attention modes.
    x_out: leave x untouched and continue the stream in x_out (aaclip_blocks_to) -- x stays valid as a tap.
    x_outs: one tensor per block = the buffer holding the stream after that block (aaclip_blocks_taps): a buffer
    that later blocks do not write again is a tap, and the whole tower is one call."""
    blocks = list(blocks)
    if not blocks:
        return
    if len({bool(getattr(b, "surgery", False)) for b in blocks}) > 1:
        raise ValueError("run_blocks: split the run where the attention mode changes")
    mode = int(causal)
    if getattr(blocks[0], "surgery", False):
        if causal:
            raise ValueError("the V-V attention block takes no mask")
        mode = ATTN_VV_BATCH
    require_gpu(x, "block")
    lib = _lib.load()
    D = x.shape[1]
    F = blocks[0].mlp.c_fc.weight.shape[0]
    arr = (BlockWeights * len(blocks))()
    refs = []
    for i, blk in enumerate(blocks):
        aw = adapter_weights[i] if adapter_weights is not None else None
        w, r = pack_block(blk, code, aw)
        arr[i] = w
        refs.append(r)
    ws = Workspace.for_rows(x.device, code, B * L, D, F, 0)
    if x_out is not None:
        require_gpu(x_out, "block")
        if x_out.shape != x.shape or x_out.dtype != torch.float32 or not x_out.is_contiguous():
            raise ValueError("x_out must be a contiguous fp32 tensor of x's shape")
    if x_outs is not None:
        if x_out is not None or len(x_outs) != len(blocks):
            raise ValueError("x_outs: one output tensor per block (and no x_out)")
        ptrs = (C.c_void_p * len(blocks))()
        for i, t in enumerate(x_outs):
            require_gpu(t, "block")
            if t.shape != x.shape or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("x_outs must be contiguous fp32 tensors of x's shape")
            ptrs[i] = t.data_ptr()
        _lib.check(lib.aaclip_blocks_taps(x.data_ptr(), ptrs, arr, len(blocks), float(mix), B, L, D, heads, F, mode,
                                          code, ws.data_ptr(), ws.numel(), _stream(x.device)), "blocks")
        del refs
        return
    dst = x if x_out is None else x_out
    _lib.check(lib.aaclip_blocks_to(x.data_ptr(), dst.data_ptr(), arr, len(blocks), float(mix), B, L, D, heads, F, mode,
                                    code, ws.data_ptr(), ws.numel(), _stream(x.device)), "blocks")
    del refs


def tap_head(x: torch.Tensor, ln_post, proj_weight: torch.Tensor, act: bool, B: int, L: int, code: int,
             det_weight: Optional[torch.Tensor] = None, keep_rows: bool = False):
    """reference model/adapter.py:171-184 -> (seg [B,L-1,E] unit rows, det [B,E] or None).
    keep_rows: also return ln_post(x) [B*L, .] in the layout of `code` (uint8 [B*L, 4D] split8 rows under fp16x2): the
    IQM branch reads the same rows (reference model/adapter.py:205-208)."""
    require_gpu(x, "tap_head")
    lib = _lib.load()
    D, E = x.shape[1], proj_weight.shape[0]
    seg = torch.empty(B, L - 1, E, dtype=torch.float32, device=x.device)
    det = torch.empty(B, E, dtype=torch.float32, device=x.device) if det_weight is not None else None
    pw = CACHE.get(proj_weight, code)
    dw = CACHE.get(det_weight, code) if det_weight is not None else None
    lw, lb = _f32c(ln_post.weight), _f32c(ln_post.bias)
    ws = Workspace.for_rows(x.device, code, B * L, D, 0, E)
    if keep_rows:
        rows = (torch.empty(B * L, 4 * D, dtype=torch.uint8, device=x.device) if code == F16X2
                else torch.empty(B * L, D, dtype=torch_dtype(code), device=x.device))
        _lib.check(lib.aaclip_tap_head_keep_rows(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), pw.data_ptr(), int(act),
                                                 seg.data_ptr(), _ptr(dw), _ptr(det), rows.data_ptr(), B, L, D, E, code,
                                                 ws.data_ptr(), ws.numel(), _stream(x.device)), "tap_head_keep_rows")
        return seg, det, rows
    _lib.check(lib.aaclip_tap_head(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), pw.data_ptr(), int(act),
                                   seg.data_ptr(), _ptr(dw), _ptr(det), B, L, D, E, code, ws.data_ptr(), ws.numel(),
                                   _stream(x.device)), "tap_head")
    return seg, det


def row_head(x: torch.Tensor, tokens: Optional[torch.Tensor], ln, proj: torch.Tensor, kind: str, act: bool, n: int,
             T: int, mode: int, code: int) -> torch.Tensor:
    """LayerNorm + row pick + projection (reference model/adapter.py:297-299,
    model/model.py:198-200, model/transformer.py:542-546).  kind: 'plain' for an
    [E,D] Linear weight, 'transpose' for a [D,E] projection parameter."""
    require_gpu(x, "row_head")
    lib = _lib.load()
    D = x.shape[1]
    pw = CACHE.get(proj, code, kind)
    E = pw.shape[0]
    out = torch.empty(n, E, dtype=torch.float32, device=x.device)
    lw, lb = _f32c(ln.weight), _f32c(ln.bias)
    ws = Workspace.for_rows(x.device, code, n * T + n, D, 0, E)
    tk = None
    if tokens is not None:
        tk = tokens.to(device=x.device, dtype=torch.int32).contiguous()
    _lib.check(lib.aaclip_row_head(x.data_ptr(), _ptr(tk), lw.data_ptr(), lb.data_ptr(), pw.data_ptr(), int(act),
                                   out.data_ptr(), n, T, D, E, mode, code, ws.data_ptr(), ws.numel(),
                                   _stream(x.device)), "row_head")
    return out


def text_embed(tokens: torch.Tensor, table: torch.Tensor, pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference model/adapter.py:277-281 -> (x [n*T, D] fp32, int32 tokens on device)."""
    lib = _lib.load()
    table = _f32c(table)
    require_gpu(table, "text_embed")
    tk = tokens.to(device=table.device, dtype=torch.int32).contiguous()
    n, T = tk.shape
    pos = _f32c(pos)
    if pos.shape[0] < T:
        raise RuntimeError("text longer than the positional embedding")
    D = table.shape[1]
    x = torch.empty(n * T, D, dtype=torch.float32, device=table.device)
    _lib.check(lib.aaclip_text_embed(tk.data_ptr(), table.data_ptr(), pos.data_ptr(), x.data_ptr(), n, T, D,
                                     table.shape[0], _stream(table.device)), "text_embed")
    return x, tk


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
              out_code: int = F32) -> torch.Tensor:
    """reference model/transformer.py:37-43 on any [..., D] tensor."""
    require_gpu(x, "layernorm")
    lib = _lib.load()
    xc = _f32c(x)
    D = xc.shape[-1]
    rows = xc.numel() // D
    oshape = xc.shape if out_code != F16X2 else (*xc.shape[:-1], 2 * D)   # split8 rows: 4 bytes per element
    out = torch.empty(oshape, dtype=_TORCH_DT[out_code], device=x.device)
    w, b = _f32c(weight), _f32c(bias)
    _lib.check(lib.aaclip_layernorm(xc.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), out_code, rows, D,
                                    float(eps), _stream(x.device)), "layernorm")
    return out


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: bool, code: int,
           kind: str = "plain") -> torch.Tensor:
    """y = act(x W^T [+ b]) through the HIP GEMM (used by SimpleAdapter/SimpleProj
    when a caller invokes those modules directly)."""
    require_gpu(x, "linear")
    lib = _lib.load()
    K = x.shape[-1]
    w = CACHE.get(weight, code, kind)
    N = w.shape[0]
    a = x.detach().reshape(-1, K)
    a = split_rows(a) if code == F16X2 else a.to(_TORCH_DT[code]).contiguous()
    out = torch.empty(a.shape[0], N, dtype=torch.float32, device=x.device)
    b = _f32c(bias) if bias is not None else None
    lda = 2 * K if code == F16X2 else K            # split8 rows: 4 bytes per element = 2K halves
    _lib.check(lib.aaclip_gemm(code, _lib.EPI_ACT_F32, a.data_ptr(), lda, w.data_ptr(), _ptr(b), out.data_ptr(), N,
                               a.shape[0], N, K, int(act), 0, 1.0, _stream(x.device)), "gemm")
    return out.reshape(*x.shape[:-1], N)


def anomaly_map(seg_tokens: Sequence[torch.Tensor], text_feature: torch.Tensor, img_size: int, ksize: int,
                sigma: float) -> torch.Tensor:
    """Fused reference forward_utils.py:196-213 (test=True) over all levels + the
    level sum of test_last.py:95-100,149 -> [B, S, S]."""
    lib = _lib.load()
    segs = [_f32c(s) for s in seg_tokens]
    require_gpu(segs[0], "anomaly_map")
    B, P, E = segs[0].shape
    g = int(round(P ** 0.5))
    if g * g != P:
        raise ValueError(f"{P} patches is not a square grid")
    tf = _f32c(text_feature).to(segs[0].device)
    if tf.dim() == 2:
        stride = 0
    elif tf.dim() == 3 and tf.shape[0] == B:
        stride = E * 2
    else:
        raise ValueError("text feature must be [E,2] or [B,E,2]")
    if tf.shape[-1] != 2 or tf.shape[-2] != E:
        raise ValueError("text feature must have shape [..., E, 2]")
    out = torch.empty(B, img_size, img_size, dtype=torch.float32, device=segs[0].device)
    ws = Workspace.get(segs[0].device, len(segs) * B * P * 4 + 256)
    arr = (C.c_void_p * len(segs))(*[s.data_ptr() for s in segs])
    _lib.check(lib.aaclip_anomaly_map(arr, len(segs), tf.data_ptr(), stride, out.data_ptr(), B, g, E, img_size,
                                      ksize, float(sigma), ws.data_ptr(), ws.numel(), _stream(out.device)),
               "anomaly_map")
    return out


def similarity_map_train(seg: torch.Tensor, text_feature: torch.Tensor, img_size: int) -> torch.Tensor:
    """reference forward_utils.py:196-216 with test=False -> [B, 2, S, S]."""
    lib = _lib.load()
    seg = _f32c(seg)
    require_gpu(seg, "similarity_map")
    B, P, E = seg.shape
    g = int(round(P ** 0.5))
    if g * g != P:
        raise ValueError(f"{P} patches is not a square grid")
    tf = _f32c(text_feature).to(seg.device)
    stride = 0 if tf.dim() == 2 else E * 2
    if tf.shape[-1] != 2:
        raise AssertionError("C == 2 expected")
    out = torch.empty(B, 2, img_size, img_size, dtype=torch.float32, device=seg.device)
    ws = Workspace.get(seg.device, 2 * B * P * 4 + 256)
    _lib.check(lib.aaclip_similarity_map_train(seg.data_ptr(), tf.data_ptr(), stride, out.data_ptr(), B, g, E,
                                               img_size, ws.data_ptr(), ws.numel(), _stream(out.device)),
               "similarity_map_train")
    return out


# ------------------------------------------------------------------------------------------------
# image pre-processing (reference dataset/__init__.py:150-161), Pillow-exact on the GPU
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resample_table(in_size: int, out_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host tables of one resize axis: bounds int32 [out,2], coefs int32 [out,ksize] (library-built)."""
    lib = _lib.load()
    k = lib.aaclip_resample_ksize(int(in_size), int(out_size))
    if k < 1:
        _lib.check(k, "resample_ksize")
    bounds = torch.empty(out_size, 2, dtype=torch.int32)
    coefs = torch.empty(out_size, k, dtype=torch.int32)
    _lib.check(lib.aaclip_resample_table(int(in_size), int(out_size), bounds.data_ptr(), coefs.data_ptr()),
               "resample_table")
    return bounds, coefs


_PRE_TABLES: Dict[tuple, tuple] = {}


def _normalise_lut(mean, std) -> torch.Tensor:
    # the fp32 operations of ToTensor (.div(255)) and Normalize (.sub_(mean).div_(std)), for every byte value
    v = torch.arange(256, dtype=torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1)
    return (v.unsqueeze(0).repeat(3, 1).sub_(m).div_(s)).contiguous()


def preprocess(src_u8: torch.Tensor, img_size: int, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """uint8 [B,Hs,Ws,3] (HWC, on the GPU) -> fp32 [B,3,S,S]: BICUBIC resize, ToTensor, Normalize."""
    require_gpu(src_u8, "preprocess")
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[-1] != 3:
        raise ValueError("preprocess expects uint8 [B, H, W, 3]")
    src_u8 = src_u8.contiguous()
    B, Hs, Ws, _ = src_u8.shape
    dev = src_u8.device
    key = (dev, Hs, Ws, img_size, tuple(mean), tuple(std))
    tabs = _PRE_TABLES.get(key)
    if tabs is None:
        hb, hk = resample_table(Ws, img_size)
        vb, vk = resample_table(Hs, img_size)
        tabs = tuple(t.to(dev) for t in (hb, hk, vb, vk, _normalise_lut(mean, std)))
        _PRE_TABLES[key] = tabs
    hb, hk, vb, vk, lut = tabs
    out = torch.empty(B, 3, img_size, img_size, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().aaclip_preprocess(src_u8.data_ptr(), B, Hs, Ws, img_size, hb.data_ptr(), hk.data_ptr(),
                                             vb.data_ptr(), vk.data_ptr(), lut.data_ptr(), out.data_ptr(),
                                             _stream(dev)), "preprocess")
    return out


# ------------------------------------------------------------------------------------------------
# IQM side branch (reference model/iqm.py, model/adapter.py:186-269, test_last.py:102-147): thin wrappers of the C ABI
def gemm(code: int, epi: int, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor,
         act: int = 0) -> torch.Tensor:
    """aaclip_gemm on prepared operands: a [M, K] and w [N, K] in the compute dtype, bias fp32 [N] or None,
    out [M, N] (16-bit for EPI_BIAS / EPI_BIAS_GELU, fp32 for EPI_ACT_F32)."""
    M = a.shape[0]
    N = w.shape[0]
    if code == F16X2:                                # split rows hold 4 bytes per element; lda / ldc count halves
        lda = a.shape[1] * a.element_size() // 2
        K = lda // 2
        ldc = out.shape[1] * out.element_size() // 2 if out.dtype != torch.float32 else out.shape[1]
    else:
        lda = K = a.shape[1]
        ldc = out.shape[1]
    _lib.check(_lib.load().aaclip_gemm(code, epi, a.data_ptr(), lda, w.data_ptr(), _ptr(bias), out.data_ptr(),
                                       ldc, M, N, K, int(act), 0, 1.0, _stream(a.device)), "gemm")
    return out


def small_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, nq: int, Lk: int, heads: int,
                    kv_code: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(hd)) v for a handful of queries (reference model/iqm.py:108-139).  q fp32 [B*nq, D];
    k, v [B*Lk, D] in the kv dtype -> fp32 [B*nq, D]."""
    D = q.shape[-1]
    hd = D // heads
    out = torch.empty(B * nq, D, dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().aaclip_small_attention(kv_code, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, nq,
                                                  Lk, heads, hd, 1.0 / (hd ** 0.5), _stream(q.device)), "small_attention")
    return out


def cross_rows(qt: torch.Tensor, x: torch.Tensor, B: int, R: int, Lk: int, x_code: int) -> torch.Tensor:
    """out[b, r] = softmax_j(qt[b, r] . x[b, j]) . x[b]: R effective queries per image over the raw rows x [B*Lk, Dk]
    (reference model/iqm.py:108-139 after folding W_k into the query and W_v behind the weighted sum; include/aaclip.h).
    qt fp32 [B*R, Dk] -> fp32 [B*R, Dk]."""
    lib = _lib.load()
    Dk = x.shape[-1]
    out = torch.empty(B * R, Dk, dtype=torch.float32, device=x.device)
    ws = Workspace.get(x.device, lib.aaclip_cross_rows_workspace_bytes(B, R, Lk, Dk) + 256)
    _lib.check(lib.aaclip_cross_rows(x_code, qt.data_ptr(), x.data_ptr(), out.data_ptr(), B, R, Lk, Dk, ws.data_ptr(),
                                     ws.numel(), _stream(x.device)), "cross_rows")
    return out


def cross_rows_levels(qt: torch.Tensor, levels, B: int, R: int, rows_per_image: int, row0: int, Lk: int,
                      Dk: int) -> torch.Tensor:
    """include/aaclip.h aaclip_cross_rows_levels.  qt fp32 [B*R, nseg*Dk]; levels = row buffers, one per segment: fp16 /
    bf16 [B*rows_per_image, Dk], or uint8 split8 rows [B*rows_per_image, 4*Dk] (their fp16 halves are read)."""
    require_gpu(qt, "cross_rows_levels")
    lib = _lib.load()
    nseg = len(levels)
    x0 = levels[0]
    if x0.dtype == torch.uint8:
        xc, ldx = _lib.F16, x0.shape[1] // 2
    else:
        xc, ldx = {torch.float16: F16, torch.bfloat16: BF16}[x0.dtype], x0.shape[1]
    for x in levels:
        if x.dtype != x0.dtype or x.shape != x0.shape or not x.is_contiguous() or x.device != qt.device:
            raise ValueError("cross_rows_levels: the level buffers must agree in dtype, shape and device")
    if qt.shape != (B * R, nseg * Dk) or qt.dtype != torch.float32 or not qt.is_contiguous():
        raise ValueError("cross_rows_levels: qt must be contiguous fp32 [B*R, nseg*Dk]")
    ptrs = (C.c_void_p * nseg)(*[x.data_ptr() for x in levels])
    out = torch.empty(B * R, nseg * Dk, dtype=torch.float32, device=qt.device)
    ws = Workspace.get(qt.device, lib.aaclip_cross_rows_levels_workspace_bytes(B, nseg, Lk, Dk) + 256)
    _lib.check(lib.aaclip_cross_rows_levels(xc, qt.data_ptr(), ptrs, nseg, out.data_ptr(), B, R, rows_per_image, row0, Lk,
                                            Dk, ldx, ws.data_ptr(), ws.numel(), _stream(qt.device)), "cross_rows_levels")
    return out


def head_expand(q: torch.Tensor, heads: int, scale: float, code: int) -> torch.Tensor:
    """q fp32 [rows, D] -> [rows*H, D] in the compute dtype: row (r, h) = q[r] * scale on head h's columns, else zero."""
    rows, D = q.shape
    out = torch.empty(rows * heads, D, dtype=_TORCH_DT[code], device=q.device)
    _lib.check(_lib.load().aaclip_head_expand(code, q.data_ptr(), out.data_ptr(), rows, heads, D, float(scale),
                                              _stream(q.device)), "head_expand")
    return out


def head_diag(full: torch.Tensor, heads: int) -> torch.Tensor:
    """full fp32 [rows*H, D] -> [rows, D]: the head-diagonal blocks (row (r, h), columns of head h)."""
    D = full.shape[-1]
    rows = full.shape[0] // heads
    out = torch.empty(rows, D, dtype=torch.float32, device=full.device)
    _lib.check(_lib.load().aaclip_head_diag(full.data_ptr(), out.data_ptr(), rows, heads, D, _stream(full.device)),
               "head_diag")
    return out


def residual_layernorm(a: torch.Tensor, b: Optional[torch.Tensor], ln, eps: float) -> torch.Tensor:
    """LayerNorm(a + b) (reference model/iqm.py:150-154); fp32 [rows, D]."""
    rows, D = a.shape
    out = torch.empty_like(a)
    w, bias = _f32c(ln.weight), _f32c(ln.bias)
    _lib.check(_lib.load().aaclip_residual_layernorm(a.data_ptr(), _ptr(b), w.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                     rows, D, float(eps), _stream(a.device)), "residual_layernorm")
    return out


def combine3(a: torch.Tensor, b: Optional[torch.Tensor], c: Optional[torch.Tensor], wa: float, wb: float,
             wc: float) -> torch.Tensor:
    out = torch.empty_like(a)
    _lib.check(_lib.load().aaclip_combine3(a.data_ptr(), _ptr(b), _ptr(c), float(wa), float(wb), float(wc), out.data_ptr(),
                                           a.numel(), _stream(a.device)), "combine3")
    return out


def linear_smallk(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out_code: int) -> torch.Tensor:
    """y = x W^T + b for in_features <= 4 -> [R, N] in the compute dtype."""
    x = _f32c(x)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    w, b = _f32c(weight), (_f32c(bias) if bias is not None else None)
    out = torch.empty(x2.shape[0], w.shape[0], dtype=_TORCH_DT[out_code], device=x.device)
    _lib.check(_lib.load().aaclip_linear_smallk(out_code, x2.data_ptr(), w.data_ptr(), _ptr(b), out.data_ptr(), x2.shape[0],
                                                w.shape[0], K, _stream(x.device)), "linear_smallk")
    return out


def drop_cls_rows(src: torch.Tensor, dst: torch.Tensor, B: int, L: int, row_off: int, code: int) -> None:
    """src [B*L, E] -> rows 1.. of every image into dst [B, rows_per_image, E] at row_off."""
    E = src.shape[-1]
    _lib.check(_lib.load().aaclip_drop_cls_rows(code, src.data_ptr(), dst.data_ptr(), B, L, E, dst.shape[1], row_off,
                                                _stream(src.device)), "drop_cls_rows")


def iqm_map(seg_tokens: Sequence[torch.Tensor], queries: torch.Tensor, img_size: int, base: Optional[torch.Tensor] = None,
            w_base: float = 0.0, w_iqm: float = 1.0) -> torch.Tensor:
    """reference test_last.py:102-147 -> [B, S, S] = w_base * base + w_iqm * sum over levels of the upsampled
    sigmoid(cos(f, q_abnormal) - cos(f, q_normal))."""
    lib = _lib.load()
    segs = [_f32c(s) for s in seg_tokens]
    require_gpu(segs[0], "iqm_map")
    B, P, E = segs[0].shape
    g = int(round(P ** 0.5))
    if g * g != P:
        raise AssertionError(f"L={P} is not a perfect square")         # reference test_last.py:125
    q = _f32c(queries)
    if q.shape != (B, 2, E):
        raise ValueError("queries must be [B, 2, E] (normal, abnormal)")
    out = torch.empty(B, img_size, img_size, dtype=torch.float32, device=segs[0].device)
    bs = _f32c(base) if base is not None else None
    if bs is not None and bs.shape != out.shape:
        raise ValueError("base map must be [B, S, S]")
    ws = Workspace.get(segs[0].device, len(segs) * B * P * 4 + 256)
    arr = (C.c_void_p * len(segs))(*[s.data_ptr() for s in segs])
    _lib.check(lib.aaclip_iqm_map(arr, len(segs), q.data_ptr(), _ptr(bs), out.data_ptr(), B, g, E, img_size, float(w_base),
                                  float(w_iqm), ws.data_ptr(), ws.numel(), _stream(out.device)), "iqm_map")
    return out
