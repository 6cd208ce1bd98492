#!/usr/bin/env python3
"""Benchmark of the AA-CLIP hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of 64 synthetic 518x518 images
per GPU, already resident in HBM.  Workload (BASELINE.json configs[1]): the
ViT-L/14-336@518 visual tower with its four tap layers
(CLIP.encode_image(image, [6,12,18,24])), 1013.6 GFLOP per image.
`--workload full` runs configs[2] instead (AdaptedCLIP.forward + anomaly map,
1041.6 GFLOP per image); its rate is also reported as `full_images_per_s`.

Images shard over ranks (weak scaling, no collective in the forward); per step one
RCCL all-gather concatenates the per-rank pooled embeddings / image scores.

The JSON line carries `roofline` for the dominant kernel (the c_fc GEMM, timed
with HIP events on the launch stream inside the timed region) and, at N=1,
`cpu_baseline`: the CPU oracle (a port of the reference's math in stock torch CPU
ops) timed on the host cores for a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "aa-clip-iqm_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

GFLOP_TOWER = 1013.6   # visual tower only, per image
GFLOP_FULL = 1041.6    # + adapters, seg/det proj, map
PEAK_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md, dense
TAGS = {0: "layernorm", 1: "qkv_gemm", 2: "attention", 3: "out_proj_gemm", 4: "c_fc_gemm", 5: "c_proj_gemm",
        6: "adapter"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--workload", default="tower", choices=["tower", "full"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a one-GPU box: AACLIP_BENCH_BACKEND=gloo puts all ranks on the visible device(s) and runs the
        # same control flow (barriers, max-over-ranks, the all-gather) without RCCL, which refuses two ranks per GPU
        backend = os.environ.get("AACLIP_BENCH_BACKEND", "nccl")
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    n_gpus = world

    from aaclip_hip import _lib, synth
    from aaclip_hip.shard import gather_rows
    from model.clip import create_model
    from model.adapter import AdaptedCLIP
    import forward_utils as FU

    lib = _lib.load()
    cfg = synth.ClipCfg()
    t0 = time.time()
    clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=args.precision, force_image_size=518)
    clip.load_state_dict(synth.synth_clip_state_dict(cfg, 111), strict=True)
    model = AdaptedCLIP(clip, relu=False)
    model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
    model.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
    model.to(dev).eval()
    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(111 + rank)
    images = torch.randn(B, 3, 518, 518, generator=gen, device=dev, dtype=torch.float32)
    anchors = torch.nn.functional.normalize(torch.randn(768, 2, generator=gen, device=dev), dim=0)
    if rank == 0:
        print(f"[bench] model + data ready in {time.time() - t0:.1f}s; B={B}/GPU, {args.precision}, "
              f"workload={args.workload}", file=sys.stderr)

    gathered = [None]

    def step_tower():
        pooled, taps = clip.encode_image(images, [6, 12, 18, 24])
        gathered[0] = gather_rows(pooled)          # ONE RCCL all-gather per step (no-op at N=1)
        return taps

    def step_full():
        seg, det, _ = model(images)
        amap = FU.calculate_anomaly_map(seg, anchors, 518, domain="Industrial")
        score = FU.image_score(det, anchors)
        gathered[0] = gather_rows(score)           # ONE RCCL all-gather per step (no-op at N=1)
        return amap

    step = step_tower if args.workload == "tower" else step_full

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        cap = 24 * args.steps + 8
        _lib.check(lib.aaclip_profile_begin(1 << 4, cap), "profile_begin")   # time every c_fc GEMM launch
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        ms = (C.c_float * cap)()
        tg = (C.c_int * cap)()
        n = lib.aaclip_profile_end(ms, tg, cap)
        fc_ms = [ms[i] for i in range(n)]

        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

        extra = {}
        if not args.no_extra:
            # (every rank runs these untimed steps: step() ends in the all-gather, a collective all ranks must enter)
            # per-kernel-class breakdown of one extra (untimed) step
            cap2 = 8 * 24 + 16
            _lib.check(lib.aaclip_profile_begin(0x7F, cap2), "profile_begin")
            step()
            torch.cuda.synchronize(dev)
            ms2, tg2 = (C.c_float * cap2)(), (C.c_int * cap2)()
            n2 = lib.aaclip_profile_end(ms2, tg2, cap2)
            br = {}
            for i in range(n2):
                br[TAGS[tg2[i]]] = br.get(TAGS[tg2[i]], 0.0) + ms2[i]
            extra["kernel_ms_per_step"] = {k: round(v, 3) for k, v in sorted(br.items())}
            # the other workload, 2 steps
            other = step_full if args.workload == "tower" else step_tower
            other()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(2):
                other()
            torch.cuda.synchronize(dev)
            key = "full_images_per_s" if args.workload == "tower" else "tower_images_per_s"
            extra[key] = round(2 * B / (time.perf_counter() - t1), 2)
        if dist is not None:
            dist.barrier()

    ms_per_step = dt / args.steps * 1e3
    value = n_gpus * B * args.steps / dt
    gflop_img = GFLOP_TOWER if args.workload == "tower" else GFLOP_FULL
    peak = PEAK_TFLOPS[args.precision]

    result = None
    if rank == 0:
        fc_flop = 2.0 * (B * cfg.tokens) * cfg.vision.mlp * cfg.vision.width   # per launch
        fc_avg = sum(fc_ms) / max(1, len(fc_ms))
        achieved = fc_flop / (fc_avg * 1e-3) / 1e12 if fc_avg > 0 else 0.0
        result = {
            "metric": "images/sec ViT-L/14-336@518, batch 64",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp16": "f16", "bf16": "bf16", "fp32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": ((f"ViT-L/14-336 visual tower @518x518 with 4 tap layers (encode_image), batch {B} per GPU"
                              if args.workload == "tower" else
                              f"full AA-CLIP visual side (adapters + seg/det heads + anomaly map), batch {B} per GPU")),
                "global_batch": n_gpus * B,
                "image": "518x518",
                "parallelism": f"dp{n_gpus}",
                "gflop_per_image": gflop_img,
            },
            "whole_path_tflops": round(value * gflop_img / 1e3, 1),
            "whole_path_frac_of_mfma_peak": round(value * gflop_img / 1e3 / (peak * n_gpus), 4),
            "roofline": {
                "kernel": "gemm16_256x_kernel<f16, EPI_BIAS_GELU> (mlp.c_fc, M=B*1370, N=4096, K=1024)",
                "bound": "mfma",
                "achieved": round(achieved, 1),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                **traffic_fields(args.precision, B),
                "launches_timed": len(fc_ms),
                "avg_launch_ms": round(fc_avg, 4),
                "flop_per_launch": fc_flop,
            },
        }
        result.update(extra)
        if n_gpus == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, args.workload)
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


def traffic_fields(precision, batch):
    """`traffic` = HBM bytes per launch of the dominant kernel from the PMC counters (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950), measured offline on this kernel and shape and committed under
    profiles/; null when the committed measurement does not match this run's configuration."""
    rel = "profiles/r01d_cfc_gemm_traffic.json"
    try:
        with open(os.path.join(REPO, rel)) as f:
            t = json.load(f)
    except OSError:
        return {"traffic": None}
    if precision != "fp16" or t.get("shape") != [batch * 1370, 4096, 1024]:
        return {"traffic": None}
    return {"traffic": round(t["traffic_bytes_per_launch"]), "traffic_unit": "bytes per launch",
            "algorithmic_bytes": t["algorithmic_bytes_per_launch"], "traffic_source": rel}


def cpu_baseline(cfg, workload):
    """The CPU oracle (port of the reference path, stock torch CPU fp32 ops) timed
    on the host cores: 1 warm-up + 2 timed single-image passes (~10-30 s)."""
    from aaclip_hip import synth
    from oracle import aaclip_oracle as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = synth.synth_clip_state_dict(cfg, 111)
    ia = synth.synth_image_adapter_state_dict(cfg, seed=111)
    img = synth.synth_images(1, 518, seed=111)

    def one():
        if workload == "tower":
            O.encode_image(img, sd, cfg.vision.heads, [6, 12, 18, 24])
        else:
            O.adapted_visual_forward(img, sd, ia, cfg.vision.heads)

    with torch.no_grad():
        one()
        t = time.perf_counter()
        reps = 2
        for _ in range(reps):
            one()
        dt = (time.perf_counter() - t) / reps
    return {"value": round(1.0 / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{reps} single-image passes of the same workload (batch 1) after 1 warm-up, "
                      f"torch {torch.__version__} CPU fp32, {cores} threads"}


if __name__ == "__main__":
    main()
