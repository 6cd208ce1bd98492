import os, sys, json, numpy as np, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
from aaclip_hip import engine, synth, _lib
from model.clip import create_model
from model.adapter import AdaptedCLIP
lib = _lib.load()
dev = torch.device("cuda:0")
g = np.load(os.path.join(REPO, "tests/golden/full4.npz"))
cfg = synth.ClipCfg()
clip = create_model("ViT-L-14-336", 518, pretrained=None, precision="fp16", force_image_size=518)
clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
model = AdaptedCLIP(clip, relu=False)
model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
model = model.to(dev).eval()
img = synth.synth_images(4, 518, seed=int(g["full4.seed"])).to(dev)
anchors = torch.from_numpy(np.load(os.path.join(REPO, "tests/golden/full.npz"))["full.anchors_bottle"]).to(dev)
ref = torch.from_numpy(g["full4.map_pre_blur_sum"]).double()
for name, v in (("fold on", 0), ("fold off", 1 << 17)):
    assert lib.aaclip_set_gemm_variant(v) == 0
    with torch.no_grad():
        seg, det, _ = model(img)
        fused = engine.anomaly_map(list(seg), anchors, 37, 1, 1.0)
    e = (fused.double().cpu() - ref).abs()
    print(f"{name}: map sum max err {e.max():.3e} rms {e.pow(2).mean().sqrt():.3e}")
lib.aaclip_set_gemm_variant(0)
