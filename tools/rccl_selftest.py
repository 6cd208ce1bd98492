#!/usr/bin/env python3
"""RCCL self-test of the collectives the hot path uses (the multi-GPU bench itself is the driver's to run):

    python tools/rccl_selftest.py              # one rank, world of one (what a 1-GPU box can do)
    python tools/rccl_selftest.py --ranks N    # N ranks, one per GPU, started by this script (bench.py's launcher)
    python tools/rccl_selftest.py --ranks 2 --backend gloo   # CPU rehearsal of the same control flow

Per rank: the process group bench.py creates (backend "nccl" = RCCL with device_id), all_gather_into_tensor of fp32
rows (rank-distinct contents, checked element by element), of the fp64 per-rank times, barrier, and
aaclip_hip.shard.gather_rows / gather_ragged_rows (ragged shard sizes) over the group.  Without torchrun the parent
starts the child ranks BEFORE touching a GPU (no exec of a GPU-initialised process) and exits with their status.
"""
import argparse
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))


def launch(n, argv):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "OMP_NUM_THREADS": str(max(1, (os.cpu_count() or 1) // n))})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc, pending = 0, list(procs)
    try:
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:        # the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def run_rank(backend):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from aaclip_hip import shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo")
    assert dist.get_world_size() == world and dist.get_backend() == backend
    x = (torch.arange(64 * 768, dtype=torch.float32, device=dev).view(64, 768) + 1e6 * rank)
    out = shard._all_gather(x, world)
    assert out.shape == (64 * world, 768)
    for r in range(world):
        want = torch.arange(64 * 768, dtype=torch.float32, device=dev).view(64, 768) + 1e6 * r
        assert torch.equal(out[64 * r: 64 * (r + 1)], want), f"rank {rank}: rows of rank {r} wrong"
    t = torch.tensor([1.25 + rank], device=dev, dtype=torch.float64)
    allt = torch.empty(world, device=dev, dtype=torch.float64)
    dist.all_gather_into_tensor(allt, t)
    assert [float(v) for v in allt.cpu()] == [1.25 + r for r in range(world)]
    dist.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert torch.equal(shard.gather_rows(x), out)
    # ragged shards: 10 rows split like the evaluation harness does
    total = 10
    b, e = shard.shard_range(total, rank, world)
    mine = torch.arange(b, e, dtype=torch.float32, device=dev).view(-1, 1).repeat(1, 3)
    full = shard.gather_ragged_rows(mine, total)
    assert torch.equal(full.cpu(), torch.arange(total, dtype=torch.float32).view(-1, 1).repeat(1, 3))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"rccl self-test ok: backend {backend}, world {world}, all_gather_into_tensor fp32/fp64, ragged gather, "
              "barrier")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.ranks > 1:
        sys.exit(launch(args.ranks, sys.argv[1:]))
    run_rank(args.backend)


if __name__ == "__main__":
    main()
