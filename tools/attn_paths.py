#!/usr/bin/env python3
"""How often does the long-sequence attention kernel redo a tile exactly?  Runs one full-size tower forward (B = 8,
synthetic weights and images, the bench's data) on the MEASUREMENT library and prints the tile-pass counters."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "aa-clip-iqm_amd"), REPO]
os.environ["AACLIP_LIB"] = os.path.join(REPO, "aa-clip-iqm_amd", "aaclip_hip", "libaaclip_hip_measure.so")
import torch
from aaclip_hip import _lib, synth
from model.clip import create_model
lib = _lib.load()
f = lib.aaclip_measure_attn_passes
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda:0")
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision="fp16", force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
clip = clip.to(dev).eval()
out = (C.c_ulonglong * 4)()
for name, img in (("randn images", torch.randn(8, 3, 518, 518, device=dev)), ("synth images", synth.synth_images(8, 518, seed=5).to(dev))):
    f(out, 1)
    with torch.no_grad():
        clip.encode_image(img, [6, 12, 18, 24])
    torch.cuda.synchronize()
    f(out, 1)
    t0, fast, redo = out[0], out[1], out[2]
    print(f"{name}: tile-0 passes {t0}, fast passes {fast}, exact redos {redo} ({100.0 * redo / max(1, fast):.2f} % of fast passes)")
