#!/usr/bin/env python3
"""Benchmark of the AA-CLIP hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of `--batch` (64) synthetic 518x518 images per GPU, already
resident in HBM.  Arithmetic mode: `--precision fp16x2` by default -- the fastest mode that meets BASELINE.json's
tolerance (1e-3 abs + 1e-2 rel vs the fp32 reference) on every output: fp16 MFMA main term + correction terms on the
e4m3 MFMAs; plain `fp16` is ~1.8x faster and up to ~3x outside the tolerance on taps and maps.  Workload (BASELINE.json configs[1]): the ViT-L/14-336@518 visual tower with its four tap layers
(CLIP.encode_image(image, [6,12,18,24])), 1013.6 GFLOP per image.  `--workload full` runs
configs[2] instead (AdaptedCLIP.forward + anomaly map, 1041.6 GFLOP per image); its rate is also reported as
`full_images_per_s`.

Ranks.  Under torchrun (WORLD_SIZE set) this process is one rank.  Without it, `--gpus N` with N > 1 makes THIS
process a launcher: it starts N child ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU)
before anything here has touched the GPU, waits for them and exits with their status; rank 0 prints the JSON line.
Images shard over ranks (weak scaling, no collective in the forward); per step ONE RCCL all-gather concatenates the
per-rank pooled embeddings / image scores.  `ranks_seen` is torch.distributed's world size, `per_rank_images_per_s`
each rank's own rate; `value` is all images of all ranks over the slowest rank's time.

The JSON line carries `roofline` for the dominant kernel (the c_fc GEMM, timed with HIP events on the launch stream
inside the timed region), `parity_vs_north_star` (where the timed arithmetic mode stands against BASELINE.json's
tolerance, from the committed GPU test record), `<mode>_companion` for the other arithmetic modes on the same workload
(fp32 = exact, the reference's type; fp16x2 = split fp16 on the 16-bit MFMAs, inside the tolerance; fp16 = fastest,
outside it on taps and maps) and, at N=1, `cpu_baseline`: the CPU oracle (a port of the reference's math in stock torch CPU ops) timed on the
host cores at the reference's 4-thread cap and on all cores, batch 1 and 2, median of 3.

`--rehearse-cpu` is a control-flow rehearsal for the CPU test of the launcher (gloo, no GPU, no kernels): it times a
stand-in step and labels its output as such; it never produces a valid metric line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "aa-clip-iqm_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GFLOP_TOWER = 1013.6   # visual tower only, per image
GFLOP_FULL = 1041.6    # + adapters, seg/det proj, map
PEAK_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3, "fp16x2": 2500.0}   # MI355X_MICROARCH.md, dense
DTYPE_NAME = {"fp16": "f16", "bf16": "bf16", "fp32": "f32", "fp16x2": "f16x2"}
# fp16x2 (split fp16, include/aaclip.h AACLIP_F16X2): a GEMM accumulates the fp16 product plus two correction products on
# the block-scaled e4m3 MFMAs (twice the fp16 rate): 2.0 fp16-MFMA time units per algorithmic unit, 1.5 where the weight
# is exact in fp16 (one correction product); attention at L >= 512: q.k^T as one fp16 product + two e4m3 correction products,
# p.v as one fp16 product = 1.5 units (csrc/attention.hip, QK8).  Rates and roofline fractions
# count the ALGORITHMIC flops (1013.6 GFLOP per image) against the fp16 MFMA peak; the MFMA pipe is busy
# `mfma_time_multiple` times as long as those flops alone would keep it.
TAGS = {0: "layernorm", 1: "qkv_gemm", 2: "attention", 3: "out_proj_gemm", 4: "c_fc_gemm", 5: "c_proj_gemm",
        6: "adapter"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    # default = the fastest arithmetic mode that meets BASELINE.json's tolerance (1e-3 abs + 1e-2 rel vs the fp32
    # reference) on every output; plain fp16 is ~2x faster and up to ~3x outside it on taps and maps (`fp16_companion`)
    ap.add_argument("--precision", default="fp16x2", choices=["fp16x2", "fp16", "bf16", "fp32"])
    ap.add_argument("--clip-weights", default="fp32", choices=["fp32", "fp16"],
                    help="synthetic CLIP weights as drawn (fp32) or rounded through fp16 like OpenAI's stored checkpoint "
                         "(fp16x2 then skips the weight-lo product of those matrices: 2 MFMA products instead of 3)")
    ap.add_argument("--workload", default="tower", choices=["tower", "full"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="control-flow rehearsal on CPU/gloo with a stand-in step (launcher test); not a measurement")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Start n child ranks of this script and wait for them.  Nothing in this process has initialised the GPU (torch
    is not even imported here), and the children are ordinary child processes -- no exec of a GPU-initialised
    process.  Children inherit stdout/stderr: rank 0 prints the JSON line."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "AACLIP_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:       # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        # N ranks share the host: without a cap each builds its model with all cores' worth of intra-op threads
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))
    import torch

    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: reporting the {world} ranks that exist",
              file=sys.stderr)
    dist = None
    rehearse = args.rehearse_cpu
    devices = 0 if rehearse else torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a one-GPU box: AACLIP_BENCH_BACKEND=gloo puts all ranks on the visible device(s) and runs the
        # same control flow (barriers, max-over-ranks, the all-gather) without RCCL, which refuses two ranks per GPU
        backend = "gloo" if rehearse else os.environ.get("AACLIP_BENCH_BACKEND", "nccl")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            local = local % max(devices, 1)
            torch.cuda.set_device(local)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
    if rehearse:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    n_gpus = world
    ranks_seen = dist.get_world_size() if dist is not None else 1

    from aaclip_hip.shard import gather_rows
    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(111 + rank)
    gathered = [None]
    lib = cfg = None
    t0 = time.time()
    if rehearse:
        x = torch.randn(B, 768, generator=gen)
        w = torch.randn(768, 768, generator=gen)

        def step():
            gathered[0] = gather_rows(x @ w)
            return gathered[0]
        step_tower = step_full = step
    else:
        from aaclip_hip import _lib, synth
        from model.clip import create_model
        from model.adapter import AdaptedCLIP
        import forward_utils as FU

        lib = _lib.load()
        cfg = synth.ClipCfg()

        def build(precision):
            clip = create_model("ViT-L-14-336", 518, pretrained=None, precision=precision, force_image_size=518)
            sd = synth.synth_clip_state_dict(cfg, 111)
            if args.clip_weights == "fp16":
                sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
            clip.load_state_dict(sd, strict=True)
            model = AdaptedCLIP(clip, relu=False)
            model.image_adapter.load_state_dict(synth.synth_image_adapter_state_dict(cfg, seed=111), strict=True)
            model.text_adapter.load_state_dict(synth.synth_text_adapter_state_dict(cfg, seed=111), strict=True)
            return clip, model.to(dev).eval()

        clip, model = build(args.precision)
        images = torch.randn(B, 3, 518, 518, generator=gen, device=dev, dtype=torch.float32)
        anchors = torch.nn.functional.normalize(torch.randn(768, 2, generator=gen, device=dev), dim=0)

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
    if rank == 0:
        print(f"[bench] model + data ready in {time.time() - t0:.1f}s; B={B}/GPU, {args.precision}, "
              f"workload={args.workload}, ranks={ranks_seen}", file=sys.stderr)

    step = step_tower if args.workload == "tower" else step_full

    def sync():
        if not rehearse:
            torch.cuda.synchronize(dev)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    fc_ms = []
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        cap = 24 * args.steps + 8
        if lib is not None:
            _lib.check(lib.aaclip_profile_begin(1 << 4, cap), "profile_begin")   # time every c_fc GEMM launch
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_own = time.perf_counter() - t0
        if lib is not None:
            ms = (C.c_float * cap)()
            tg = (C.c_int * cap)()
            n = lib.aaclip_profile_end(ms, tg, cap)
            fc_ms = [ms[i] for i in range(n)]
        rows_gathered = int(gathered[0].shape[0]) if torch.is_tensor(gathered[0]) else None

        # max over ranks (the timed region ends when the slowest rank ends) + every rank's own time
        times = torch.tensor([dt_own], device=dev, dtype=torch.float64)
        if dist is not None:
            allt = torch.empty(ranks_seen, device=dev, dtype=torch.float64)
            dist.all_gather_into_tensor(allt, times)
            per_rank_dt = [float(v) for v in allt.cpu()]
        else:
            per_rank_dt = [dt_own]
        dt = max(per_rank_dt)

        extra = {}
        if not args.no_extra and not rehearse:
            # (every rank runs these untimed steps: step() ends in the all-gather, a collective all ranks must enter)
            # per-kernel-class breakdown of one extra (untimed) step
            cap2 = 8 * 24 + 16
            _lib.check(lib.aaclip_profile_begin(0x7F, cap2), "profile_begin")
            step()
            sync()
            ms2, tg2 = (C.c_float * cap2)(), (C.c_int * cap2)()
            n2 = lib.aaclip_profile_end(ms2, tg2, cap2)
            br = {}
            for i in range(n2):
                br[TAGS[tg2[i]]] = br.get(TAGS[tg2[i]], 0.0) + ms2[i]
            extra["kernel_ms_per_step"] = {k: round(v, 3) for k, v in sorted(br.items())}
            # the other workload, 2 steps
            other = step_full if args.workload == "tower" else step_tower
            other()
            sync()
            t1 = time.perf_counter()
            for _ in range(2):
                other()
            sync()
            key = "full_images_per_s" if args.workload == "tower" else "tower_images_per_s"
            extra[key] = round(2 * B / (time.perf_counter() - t1), 2)
            if n_gpus == 1:
                # the other arithmetic modes on the same workload and batch, so that one line shows all three:
                # fp32 (the reference's type, exact), fp16x2 (16-bit MFMAs inside the north-star tolerance), fp16
                for other_p in ("fp32", "fp16x2", "fp16"):
                    if other_p == args.precision or (args.precision == "bf16" and other_p == "fp16"):
                        continue
                    try:    # a secondary measurement must never cost the headline line (single rank: nothing to hang)
                        extra[f"{other_p}_companion"] = companion(other_p, build, args, B, dev, torch)
                    except Exception as e:   # noqa: BLE001
                        extra[f"{other_p}_companion"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                if args.precision == "fp16x2" and args.clip_weights == "fp32":
                    # the deployment case: CLIP weights exact in fp16 (OpenAI's checkpoint is stored in fp16), for which
                    # the weight-lo correction product is skipped
                    try:
                        args.clip_weights = "fp16"
                        extra["fp16x2_fp16_exact_clip_weights"] = companion("fp16x2", build, args, B, dev, torch)
                    except Exception as e:   # noqa: BLE001
                        extra["fp16x2_fp16_exact_clip_weights"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    finally:
                        args.clip_weights = "fp32"
        if dist is not None:
            dist.barrier()

    ms_per_step = dt / args.steps * 1e3
    value = sum(B * args.steps for _ in per_rank_dt) / dt
    gflop_img = GFLOP_TOWER if args.workload == "tower" else GFLOP_FULL
    peak = PEAK_TFLOPS[args.precision]

    if rank == 0:
        result = {
            "metric": (f"images/sec ViT-L/14-336@518, batch {B}" if not rehearse else
                       "REHEARSAL of the launcher / collective control flow on CPU (gloo): no GPU work, not a measurement"),
            "value": round(value, 2) if not rehearse else None,
            "unit": "images/s",
            "n_gpus": n_gpus,
            "ranks_seen": ranks_seen,
            "devices_visible": devices,
            "threads_per_rank": torch.get_num_threads(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision],
            "data": "synthetic" if not rehearse else "rehearsal",
            "config": {
                "workload": ((f"ViT-L/14-336 visual tower @518x518 with 4 tap layers (encode_image), batch {B} per GPU"
                              if args.workload == "tower" else
                              f"full AA-CLIP visual side (adapters + seg/det heads + anomaly map), batch {B} per GPU")),
                "global_batch": n_gpus * B,
                "image": "518x518",
                "parallelism": f"dp{n_gpus}",
                "gflop_per_image": gflop_img,
                "clip_weights": ("random fp32" if args.clip_weights == "fp32" else
                                 "random, rounded through fp16 (exact in fp16, like OpenAI's stored checkpoint)"),
            },
            "per_rank_images_per_s": [round(B * args.steps / t, 2) for t in per_rank_dt],
            "rows_all_gathered_per_step": rows_gathered,
            "launch": ("self-launched child ranks" if os.environ.get("AACLIP_BENCH_SELF_LAUNCHED") else
                       ("external launcher (WORLD_SIZE set)" if "WORLD_SIZE" in os.environ else "single process")),
        }
        if n_gpus > 1 and devices and devices < n_gpus:
            result["note"] = (f"{n_gpus} ranks share {devices} visible device(s) (gloo rehearsal): the sum is not a "
                              "multi-GPU throughput")
        if not rehearse:
            fc_flop = 2.0 * (B * cfg.tokens) * cfg.vision.mlp * cfg.vision.width   # per launch
            fc_avg = sum(fc_ms) / max(1, len(fc_ms))
            achieved = fc_flop / (fc_avg * 1e-3) / 1e12 if fc_avg > 0 else 0.0
            tname = DTYPE_NAME[args.precision]
            kname = {"fp32": "gemm32_kernel<EPI_BIAS_GELU>",
                     "fp16x2": "gemm16_256x_kernel<f16, EPI_BIAS_GELU, NP=%d>" % (
                         3 if args.clip_weights == "fp16" else 4)}.get(
                             args.precision, f"gemm16_256x_kernel<{tname}, EPI_BIAS_GELU>")
            result.update({
                "whole_path_tflops": round(value * gflop_img / 1e3, 1),
                "whole_path_frac_of_mfma_peak": round(value * gflop_img / 1e3 / (peak * n_gpus), 4),
                # share of the fp16 MFMA pipe's time the path keeps busy = algorithmic fraction x the mode's MFMA time
                # multiple (fp16x2: GEMMs 82 % of the flops at 2.0 -- 1.5 with fp16-exact weights --, attention 18 % at 1.5)
                "whole_path_mfma_time_frac": round(value * gflop_img / 1e3 / (peak * n_gpus) * (
                    ((0.82 * (1.5 if args.clip_weights == "fp16" else 2.0) + 0.18 * 1.5) if args.precision == "fp16x2"
                     else 1.0)), 4),
                "roofline": {
                    "kernel": kname + f" (mlp.c_fc, M={B}*1370, N=4096, K=1024)",
                    "bound": "mfma",
                    "achieved": round(achieved, 1),
                    "peak": peak,
                    "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4),
                    **traffic_fields(args.precision, B, args.clip_weights),
                    "launches_timed": len(fc_ms),
                    "avg_launch_ms": round(fc_avg, 4),
                    "flop_per_launch": fc_flop,
                    "mfma_time_multiple": (1.5 if args.clip_weights == "fp16" else 2.0) if args.precision == "fp16x2" else 1.0,
                },
                "parity_vs_north_star": parity_fields(args.precision),
            })
        result.update(extra)
        if n_gpus == 1 and not args.no_cpu_baseline and not rehearse:
            try:
                result["cpu_baseline"] = cpu_baseline(cfg, args.workload)
            except Exception as e:   # noqa: BLE001
                result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


def parity_fields(precision):
    """Where this arithmetic mode stands against BASELINE.json's tolerance (|a - b| <= 1e-3 + 1e-2 |b| vs the fp32
    reference) on the full-size B = 4 golden record: the largest error-to-bound ratio over raw taps, pooled embedding,
    per-level and summed pre-blur maps, as measured by tests/test_gpu_configs.py on MI355X and committed under
    profiles/ (> 1 means OUTSIDE the tolerance).  The timed path is bit-identical to that B = 4 run per image."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_parity_errors.json")), reverse=True):
        try:
            with open(path) as f:
                e = json.load(f)
        except (OSError, ValueError):
            continue
        keys = [k for k in e if k.startswith(precision + ".b4.") and ("tap" in k or "map_pre_blur" in k or "pooled" in k)]
        feats = [k for k in e if k.startswith(precision + ".b4.") and ("seg" in k or k.endswith(".det"))]
        if not keys:
            continue
        worst = max(keys, key=lambda k: e[k]["max_ratio_to_north_star"])
        return {"taps_and_maps_max_ratio": round(e[worst]["max_ratio_to_north_star"], 3), "worst_output": worst,
                "features_max_ratio": round(max(e[k]["max_ratio_to_north_star"] for k in feats), 3) if feats else None,
                "inside_north_star": e[worst]["max_ratio_to_north_star"] <= 1.0,
                "source": os.path.relpath(path, REPO)}
    return None


def companion(precision, build, args, B, dev, torch):
    """The same workload and batch in another arithmetic mode (fp32 = v_mfma_f32_32x32x2_f32, the reference's type,
    model/clip.py:88; fp16x2 = split fp16; fp16): 1 warm-up + 2 timed steps, images/s and the fraction of that
    mode's MFMA peak (algorithmic flops)."""
    clip32, model32 = build(precision)
    import forward_utils as FU
    gen = torch.Generator(device=dev)
    gen.manual_seed(112)
    images = torch.randn(B, 3, 518, 518, generator=gen, device=dev, dtype=torch.float32)
    anchors = torch.nn.functional.normalize(torch.randn(768, 2, generator=gen, device=dev), dim=0)

    def step():
        if args.workload == "tower":
            return clip32.encode_image(images, [6, 12, 18, 24])
        seg, det, _ = model32(images)
        return FU.calculate_anomaly_map(seg, anchors, 518, domain="Industrial"), FU.image_score(det, anchors)

    step()
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    steps = 2
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t
    gflop = GFLOP_TOWER if args.workload == "tower" else GFLOP_FULL
    rate = steps * B / dt
    del clip32, model32
    torch.cuda.empty_cache()
    return {"dtype": DTYPE_NAME[precision], "value": round(rate, 2), "unit": "images/s", "steps": steps, "batch": B,
            "ms_per_step": round(dt / steps * 1e3, 2), "whole_path_tflops": round(rate * gflop / 1e3, 1),
            "frac_of_mfma_peak": round(rate * gflop / 1e3 / PEAK_TFLOPS[precision], 4),
            "peak_tflops": PEAK_TFLOPS[precision], "parity_vs_north_star": parity_fields(precision)}


def traffic_fields(precision, batch, clip_weights="fp32"):
    """`traffic` = HBM bytes per launch of the dominant kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), measured offline
    on this kernel and shape and committed under profiles/.  A committed measurement only counts for the kernel it was
    taken on: the file records the sha256 of the kernel's sources (`kernel_revision`, aaclip_hip/_lib.py
    kernel_source_revision) and a file whose revision differs from the sources of this run is refused (null)."""
    import glob
    from aaclip_hip import _lib
    rev = _lib.kernel_source_revision()
    cands = sorted(glob.glob(os.path.join(REPO, "profiles", "*cfc_gemm_traffic*.json")), reverse=True)
    stale = None
    for path in cands:
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if (t.get("precision", "fp16") != precision or t.get("shape") != [batch * 1370, 4096, 1024]
                or (precision == "fp16x2" and t.get("clip_weights", "fp32") != clip_weights)):
            continue
        rel = os.path.relpath(path, REPO)
        if t.get("kernel_revision") != rev:
            stale = stale or rel
            continue
        return {"traffic": round(t["traffic_bytes_per_launch"]), "traffic_unit": "bytes per launch",
                "algorithmic_bytes": t["algorithmic_bytes_per_launch"], "traffic_source": rel,
                "kernel_revision": rev}
    out = {"traffic": None, "kernel_revision": rev}
    if stale:
        out["traffic_note"] = f"{stale} was measured on another revision of the kernel sources: refused"
    return out


def cpu_baseline(cfg, workload):
    """The CPU oracle (port of the reference path, stock torch CPU fp32 ops) timed on the host cores of this box, as
    SURVEY.md 8(d) specifies: at the reference's hard-coded 4-thread cap (test_last.py:28-35) and on all cores (capped
    at 32), batch 1 and batch 2, one warm-up per thread setting, median of 3.  `value` is the best all-core rate."""
    import statistics
    import torch
    from aaclip_hip import synth
    from oracle import aaclip_oracle as O
    sd = synth.synth_clip_state_dict(cfg, 111)
    ia = synth.synth_image_adapter_state_dict(cfg, seed=111)
    imgs = synth.synth_images(2, 518, seed=111)
    all_cores = min(os.cpu_count() or 1, 32)

    def one(img):
        if workload == "tower":
            O.encode_image(img, sd, cfg.vision.heads, [6, 12, 18, 24])
        else:
            O.adapted_visual_forward(img, sd, ia, cfg.vision.heads)

    runs = []
    t_start = time.perf_counter()
    with torch.no_grad():
        for threads in sorted({min(4, all_cores), all_cores}):
            torch.set_num_threads(threads)
            one(imgs[:1])                                     # warm-up
            for b in (1, 2):
                ts = []
                for _ in range(3):
                    t = time.perf_counter()
                    one(imgs[:b])
                    ts.append(time.perf_counter() - t)
                med = statistics.median(ts)
                runs.append({"threads": threads, "batch": b, "median_s": round(med, 3),
                             "images_per_s": round(b / med, 3)})
    best = max((r for r in runs if r["threads"] == all_cores), key=lambda r: r["images_per_s"])
    return {"value": best["images_per_s"], "unit": "images/s", "cores": all_cores, "cores_available": os.cpu_count(),
            "kind": "port",
            "sample": (f"oracle/aaclip_oracle.py, same workload, batch 1 and 2, 1 warm-up + median of 3 per setting, "
                       f"torch {torch.__version__} CPU fp32; {time.perf_counter() - t_start:.0f} s of CPU work; "
                       f"os.cpu_count() = {os.cpu_count()}"),
            "reference_thread_cap_4": [r for r in runs if r["threads"] == min(4, all_cores)],
            "all_cores": [r for r in runs if r["threads"] == all_cores]}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
