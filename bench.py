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
    ap.add_argument("--feed", default="resident", choices=["resident", "pinned"],
                    help="resident: the batch stays in HBM (the metric's definition).  pinned: every step computes on a "
                         "batch that was copied from pinned host memory on a side stream while the previous step ran "
                         "(two device buffers, one event each way; reference test_last.py:70-71 feeds from a DataLoader)")
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
# overlapped input feed (reference test_last.py:70-71: every batch arrives from the host)
# ----------------------------------------------------------------------------------------------------------------
class PinnedFeed:
    """Double-buffered host -> device feed.  Two pinned host batches stand for what a DataLoader with pin_memory hands
    over; batch i is copied into device buffer i & 1 on a side stream while batch i - 1 computes on the other buffer:
      copy stream:     wait free[k] -> H2D copy -> record ready[k]
      compute stream:  wait ready[k] -> step on buffer k -> record free[k] -> start the copy of batch i + 2
    On CPU (the gloo rehearsal of the launcher) there are no streams: the same calls in the same order, plain copies."""

    def __init__(self, torch, dev, shape, seed, rehearse=False):
        self.torch, self.dev, self.rehearse = torch, dev, rehearse
        g = torch.Generator()
        g.manual_seed(seed)
        self.host = [torch.randn(*shape, generator=g) for _ in range(2)]
        self.dbuf = [torch.empty(*shape, device=dev) for _ in range(2)]
        self.issued = 0        # batches whose copy has been started
        self.taken = 0         # batches handed to the compute stream
        if not rehearse:
            self.host = [h.pin_memory() for h in self.host]
            self.copy_stream = torch.cuda.Stream(dev)
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            for e in self.free:
                e.record(torch.cuda.current_stream(dev))
        self._start_copy()
        self._start_copy()

    def _start_copy(self):
        k = self.issued & 1
        if self.rehearse:
            self.dbuf[k].copy_(self.host[k])
        else:
            with self.torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.free[k])
                self.dbuf[k].copy_(self.host[k], non_blocking=True)
                self.ready[k].record(self.copy_stream)
        self.issued += 1

    def next(self):
        """The next batch, valid on the current stream."""
        k = self.taken & 1
        if not self.rehearse:
            self.torch.cuda.current_stream(self.dev).wait_event(self.ready[k])
        self.taken += 1
        return self.dbuf[k]

    def done(self):
        """The step on the batch last handed out has been queued: its buffer may be refilled once that step is over."""
        k = (self.taken - 1) & 1
        if not self.rehearse:
            self.free[k].record(self.torch.cuda.current_stream(self.dev))
        self._start_copy()


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

        cur = {"images": x}

        def step():
            gathered[0] = gather_rows(cur["images"] @ w)
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
        cur = {"images": images}

        def step_tower():
            pooled, taps = clip.encode_image(cur["images"], [6, 12, 18, 24])
            gathered[0] = gather_rows(pooled)          # ONE RCCL all-gather per step (no-op at N=1)
            return taps

        def step_full():
            seg, det, _ = model(cur["images"])
            amap = FU.calculate_anomaly_map(seg, anchors, 518, domain="Industrial")
            score = FU.image_score(det, anchors)
            gathered[0] = gather_rows(score)           # ONE RCCL all-gather per step (no-op at N=1)
            return amap
    if rank == 0:
        print(f"[bench] model + data ready in {time.time() - t0:.1f}s; B={B}/GPU, {args.precision}, "
              f"workload={args.workload}, ranks={ranks_seen}", file=sys.stderr)

    step_resident = step_tower if args.workload == "tower" else step_full

    def make_fed_step(inner):
        """`inner` on batches that arrive from pinned host memory, copy overlapped with the previous step"""
        feed = PinnedFeed(torch, dev, tuple(cur["images"].shape), 977 + rank, rehearse)

        def fed():
            cur["images"] = feed.next()
            out = inner()
            feed.done()
            return out
        return fed

    step = make_fed_step(step_resident) if args.feed == "pinned" else step_resident

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
            if args.feed == "resident":
                # the same step on batches that arrive from pinned host memory (copy on a side stream under the
                # previous step), next to the resident-data rate of the same process: what the H2D feed costs
                fed = make_fed_step(step_resident)
                nf = max(3, min(args.steps, 10))
                for _ in range(2):
                    fed()
                fence()
                t1 = time.perf_counter()
                for _ in range(nf):
                    fed()
                fence()
                dt_fed = time.perf_counter() - t1
                for _ in range(2):
                    step_resident()
                fence()
                t1 = time.perf_counter()
                for _ in range(nf):
                    step_resident()
                fence()
                dt_res = time.perf_counter() - t1
                extra["input_feed"] = {
                    "mode": "pinned host batches, H2D copy on a side stream overlapped with the previous step "
                            "(two device buffers, one event each way)",
                    "steps": nf, "fed_images_per_s_this_rank": round(nf * B / dt_fed, 2),
                    "resident_images_per_s_this_rank": round(nf * B / dt_res, 2),
                    "loss_pct": round(100.0 * (dt_fed - dt_res) / dt_res, 2),
                    "bytes_per_step_per_rank": B * 3 * 518 * 518 * 4}
                cur["images"] = images
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
        # parity of the timed model, measured now (rank 0; the other ranks wait in the barrier below, so that no rank tears
        # the process group down while rank 0 still works)
        parity = None
        if rank == 0 and not rehearse:
            try:
                parity = parity_in_run(clip, model, args.precision, args.clip_weights, dev, torch)
            except Exception as e:   # noqa: BLE001
                parity = {"error": f"{type(e).__name__}: {e}"[:300]}
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
                "input": ("batch resident in HBM" if args.feed == "resident" else
                          "every batch copied from pinned host memory on a side stream under the previous step"),
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
                     "fp16x2": "gemm16_256x_kernel<f16, EPI_BIAS_GELU, NP=%d, WALK>" % (
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
                "parity_vs_north_star": parity,
            })
        result.update(extra)
        if n_gpus == 1 and not args.no_extra and not rehearse:
            try:    # the text side (reported separately, never part of images/s)
                result["text_anchors"] = text_anchors_leg(model, lib, dev, torch, cfg, args.precision,
                                                          not args.no_cpu_baseline)
            except Exception as e:   # noqa: BLE001
                result["text_anchors"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if n_gpus == 1 and not args.no_cpu_baseline and not rehearse:
            try:
                result["cpu_baseline"] = cpu_baseline(cfg, args.workload)
            except Exception as e:   # noqa: BLE001
                result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


def parity_in_run(clip, model, precision, clip_weights, dev, torch):
    """Where the arithmetic mode being timed stands against BASELINE.json's tolerance (|a - b| <= 1e-3 + 1e-2 |b| vs
    the fp32 reference), MEASURED IN THIS RUN: the four images of the reference's full-size B = 4 golden record
    (tests/golden/full4.npz; full4h.npz when the CLIP weights are the fp16-exact ones) go through the model objects
    that were just timed, and the largest error-to-bound ratio is taken over raw taps + pooled embedding
    (`encode_image`) and per-level + summed pre-blur maps (`AdaptedCLIP.forward` + the map kernel); unit seg tokens and
    the det token are reported beside it.  > 1 means OUTSIDE the tolerance.  A few seconds; the goldens are data the
    reference produced (tests/golden/make_golden_full4*.py), nothing of the reference runs here."""
    import numpy as np
    from aaclip_hip import engine, synth
    gold = os.path.join(REPO, "tests", "golden")
    name = "full4h" if clip_weights == "fp16" else "full4"
    try:
        g = np.load(os.path.join(gold, name + ".npz"))
        anchors = torch.from_numpy(np.load(os.path.join(gold, "full.npz"))["full.anchors_bottle"]).to(dev)
    except OSError as e:
        return {"error": f"golden record missing: {e}"[:200]}
    img = synth.synth_images(4, 518, seed=int(g[name + ".seed"])).to(dev)
    ratios = {}

    def ratio(key, got, ref):
        got, ref = got.detach().double().cpu().reshape(-1), ref.double().reshape(-1)
        ok = bool(torch.isfinite(got).all())
        ratios[key] = float(((got - ref).abs() / (1e-3 + 1e-2 * ref.abs())).max()) if ok else float("inf")

    def sampled(key, t):
        return t.detach().reshape(-1).cpu()[torch.from_numpy(g[key + ".idx"])], torch.from_numpy(g[key + ".val"])

    with torch.no_grad():
        seg, det, _ = model(img)
        for i in range(4):
            raw = engine.anomaly_map([seg[i]], anchors, 37, 1, 1.0)
            ratio(f"map_pre_blur{i}", raw, torch.from_numpy(g[f"{name}.map_pre_blur{i}"]))
        fused = engine.anomaly_map(list(seg), anchors, 37, 1, 1.0)
        ratio("map_pre_blur_sum", fused, torch.from_numpy(g[f"{name}.map_pre_blur_sum"]))
        feats = {}
        for i in range(4):
            a, b = sampled(f"{name}.seg{i}", seg[i])
            feats[f"seg{i}"] = float(((a.double() - b.double()).abs() / (1e-3 + 1e-2 * b.double().abs())).max())
        d_ref = torch.from_numpy(g[f"{name}.det"]).double()
        feats["det"] = float(((det.double().cpu() - d_ref).abs() / (1e-3 + 1e-2 * d_ref.abs())).max())
        if name == "full4":   # the fp16-exact record holds no taps
            pooled, taps = clip.encode_image(img, [6, 12, 18, 24])
            for k, t in zip((6, 12, 18, 24), taps):
                a, b = sampled(f"full4.tap{k}", t)
                ratio(f"tap{k}", a, b)
            ratio("pooled", pooled, torch.from_numpy(g["full4.pooled"]))
    worst = max(ratios, key=ratios.get)
    return {"taps_and_maps_max_ratio": round(ratios[worst], 3), "worst_output": f"{precision}.b4.{worst}",
            "features_max_ratio": round(max(feats.values()), 3),
            "inside_north_star": max(max(ratios.values()), max(feats.values())) <= 1.0,
            "per_output_ratio": {k: round(v, 3) for k, v in sorted(ratios.items())},
            "source": f"measured in this run: B = 4 golden images through the timed model vs tests/golden/{name}.npz "
                      "(reference outputs)"}


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
    out = _companion_fields(precision, rate, steps, B, dt, gflop, clip32, model32, args, dev, torch)
    del clip32, model32
    torch.cuda.empty_cache()
    return out


def _companion_fields(precision, rate, steps, B, dt, gflop, clip32, model32, args, dev, torch):
    return {"dtype": DTYPE_NAME[precision], "value": round(rate, 2), "unit": "images/s", "steps": steps, "batch": B,
            "ms_per_step": round(dt / steps * 1e3, 2), "whole_path_tflops": round(rate * gflop / 1e3, 1),
            "frac_of_mfma_peak": round(rate * gflop / 1e3 / PEAK_TFLOPS[precision], 4),
            "peak_tflops": PEAK_TFLOPS[precision],
            "parity_vs_north_star": parity_in_run(clip32, model32, precision, args.clip_weights, dev, torch)}


GFLOP_SENTENCE = 13.57   # adapted text tower, per 77-token sentence


def text_anchors_leg(model, lib, dev, torch, cfg, precision, with_cpu):
    """The text side of the path, timed on its own ("report separately, exclude from img/s"): all 240 MVTec
    prompt sentences (15 classes x 16; reference forward_utils.py:138-192) through
    forward_utils.get_adapted_text_embedding -- ONE batched AdaptedCLIP.encode_text call (M = 240 x 77 = 18 480 rows, 12
    causal blocks of width 768 with the text adapters, reference model/adapter.py:273-304) plus the segmented means --
    in the arithmetic mode being benchmarked.  `call_ms` is the whole call (host tokenisation from the prompt table, the
    kernels, the 30 anchor means) with the device synchronised on both sides, median of 5 after one warm-up;
    `kernel_ms` the HIP-event time of the library's kernels by class for one call.  Next to it the CPU oracle's
    `adapted_encode_text` on the host cores for ONE class (16 sentences), which is how the reference runs it
    (one class at a time, forward_utils.py:185-192)."""
    import statistics
    import forward_utils as FU
    from aaclip_hip import _lib, synth
    from dataset.constants import CLASS_NAMES
    names = list(CLASS_NAMES["MVTec"])
    n_sent = 16 * len(names)

    def call():
        return FU.get_adapted_text_embedding(model, "MVTec", dev)

    with torch.no_grad():
        anchors = call()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            call()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t)
        cap = 12 * 12 + 64
        _lib.check(lib.aaclip_profile_begin(0x7F, cap), "profile_begin")
        call()
        torch.cuda.synchronize(dev)
        ms, tg = (C.c_float * cap)(), (C.c_int * cap)()
        n = lib.aaclip_profile_end(ms, tg, cap)
    br = {}
    for i in range(n):
        br[TAGS[tg[i]]] = br.get(TAGS[tg[i]], 0.0) + ms[i]
    med = statistics.median(ts)
    kernel_total = sum(br.values())
    ok = all(bool(torch.isfinite(a).all()) and tuple(a.shape) == (768, 2) for a in anchors.values())
    out = {"sentences": n_sent, "classes": len(names), "rows": n_sent * 77, "dtype": DTYPE_NAME[precision],
           "call_ms": round(med * 1e3, 3), "sentences_per_s": round(n_sent / med, 1),
           "tflops": round(n_sent * GFLOP_SENTENCE / med / 1e3, 1),
           "kernel_ms": {k: round(v, 3) for k, v in sorted(br.items())}, "kernel_ms_total": round(kernel_total, 3),
           "kernel_tflops": round(n_sent * GFLOP_SENTENCE / (kernel_total * 1e-3) / 1e3, 1) if kernel_total > 0 else None,
           "gflop_per_sentence": GFLOP_SENTENCE, "anchors_finite": ok,
           "kernels": "M = 18 480 rows: the 256-tile GEMMs of the visual tower (K = 768 / 3072), the 128-query "
                      "causal attention kernel (L = 77), LayerNorm / adapter-mix row kernels"}
    if with_cpu:
        from oracle import aaclip_oracle as O
        from model.tokenizer import tokenize
        sd = synth.synth_clip_state_dict(cfg, 111)
        ta = synth.synth_text_adapter_state_dict(cfg, seed=111)
        sentences = [s_ for grp in FU.class_sentences("MVTec", names[0]) for s_ in grp]
        tok = tokenize(sentences)
        all_cores = min(os.cpu_count() or 1, 32)
        runs = []
        with torch.no_grad():
            for threads in sorted({min(4, all_cores), all_cores}):
                torch.set_num_threads(threads)
                O.adapted_encode_text(tok, sd, ta, cfg.text.heads)
                t3 = []
                for _ in range(3):
                    t = time.perf_counter()
                    O.adapted_encode_text(tok, sd, ta, cfg.text.heads)
                    t3.append(time.perf_counter() - t)
                m3 = statistics.median(t3)
                runs.append({"threads": threads, "s_per_class": round(m3, 3),
                             "sentences_per_s": round(len(sentences) / m3, 2)})
        out["cpu_baseline"] = {"kind": "port", "unit": "sentences/s", "value": runs[-1]["sentences_per_s"],
                               "cores": all_cores, "runs": runs,
                               "sample": "oracle adapted_encode_text, the 16 sentences of one class, 1 warm-up + median "
                                         "of 3 per thread setting"}
    return out


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
