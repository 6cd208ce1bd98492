#!/usr/bin/env python3
"""Where a walked tile's time goes (gemm16_256x_kernel<..., WALK>), from s_memtime stamps per wave.
Needs the measurement library built with the stamps:  make -C aa-clip-iqm_amd/csrc measure MEASURE_DEFS=-DX_WALK_STAMP
  python tools/walk_stamps.py            # the tower's four split-fp16 shapes at B = 64"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
os.environ.setdefault("AACLIP_LIB", os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so"))
import torch
from aaclip_hip import _lib
lib = _lib.load()
lib.aaclip_debug_gemm_stamps.restype = C.c_int
lib.aaclip_debug_gemm_stamps.argtypes = [C.POINTER(C.c_double), C.c_int]
dev = torch.device("cuda:0")
M = 64 * 1370
st = torch.cuda.current_stream(dev).cuda_stream
assert lib.aaclip_set_gemm_variant(82) == 0
for name, (K, N, epi) in {"qkv": (1024, 3072, 0), "out_proj": (1024, 1024, 2), "c_fc": (1024, 4096, 1), "c_proj": (4096, 1024, 2)}.items():
    A = torch.randint(0, 255, (M, 4 * K), dtype=torch.uint8, device=dev)
    A[:, 1::2][:, :K] &= 0x3B
    W = torch.randint(0, 255, (N, 4 * K), dtype=torch.uint8, device=dev)
    W[:, 1::2][:, :K] &= 0x3B
    bias = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev) if epi == 2 else torch.empty(M, 4 * N, dtype=torch.uint8, device=dev)
    ldc = N if epi == 2 else 2 * N
    e = (C.c_double * 8)()
    for _ in range(4):
        lib.aaclip_debug_gemm_stamps(e, -2)      # (reset: the last launch's epilogue segments are read below)
        _lib.check(lib.aaclip_gemm(_lib.F16X2, epi, A.data_ptr(), 2 * K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), ldc,
                                   M, N, K, 0, 0, 1.0, st), "gemm")
    torch.cuda.synchronize()
    o = (C.c_double * 6)()
    lib.aaclip_debug_gemm_stamps(o, 2048)     # mean over the 256 x 8 waves: sums over the tiles a wave walked
    wait, loop, pre, epi_t, tiles, first = (o[i] for i in range(6))
    # s_memtime counts shader-clock cycles here (the phase times of tools/gemm_stamps.py agree with the MFMA issue rate);
    # microseconds below assume 1.9 GHz, the clock these kernels hold under the power cap
    ns = 1.0 / 1.9
    print(f"{name:9s} K={K} N={N}: {tiles:.2f} tiles per workgroup ({loop / tiles:.0f} / {pre / tiles:.0f} / {epi_t / tiles:.0f} / {wait / max(tiles - 1, 1):.0f} cycles); per tile: K loop {loop / tiles * ns / 1e3:7.2f} us, "
          f"issue of the next tile's first K tile {pre / tiles * ns / 1e3:5.2f} us, epilogue (to its last store ISSUED) "
          f"{epi_t / tiles * ns / 1e3:6.2f} us, then barrier + B0 issue + wait until the prefetched pieces AND the stores are in "
          f"{wait / max(tiles - 1, 1) * ns / 1e3:5.2f} us; first tile: entry -> K loop {first * ns / 1e3:5.2f} us", flush=True)
    torch.cuda.synchronize()
    lib.aaclip_debug_gemm_stamps(e, -2)
    if e[2] > 0:     # 16-bit outputs: four passes of 32 rows (convert + stage | read back + store)
        print(f"          epilogue of wave 0 per tile: convert + stage {e[0] / e[2]:.0f}, read back + issue the stores {e[1] / e[2]:.0f} units", flush=True)
    if e[6] > 0:     # fp32 outputs: four row groups (request the residual rows + stage | wait for them (+ earlier stores) | add + store)
        g = e[6] / 4
        print(f"          epilogue of wave 0 per tile: request rows + stage {e[3] / g:.0f}, wait for the rows (s_waitcnt vmcnt(0): also the "
              f"previous group's stores) {e[4] / g:.0f}, add + issue the stores {e[5] / g:.0f} units", flush=True)
lib.aaclip_set_gemm_variant(0)
