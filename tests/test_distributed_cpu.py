"""World-size-2 gloo test of the data-parallel sharding / single all-gather (CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aaclip_hip.shard import gather_ragged_rows, gather_rows, shard_range


def test_shard_range_partitions():
    for total in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        score = lambda i: torch.tensor([i * 0.5, i * i * 1.0])  # per-image "result" depends on the global index only
        b, e = shard_range(total, rank, world)
        local = torch.stack([score(i) for i in range(b, e)]) if e > b else torch.zeros(0, 2)
        full = gather_ragged_rows(local, total)
        ref = torch.stack([score(i) for i in range(total)])
        assert torch.equal(full, ref), (rank, full, ref)
        even = gather_rows(torch.full((3, 2), float(rank)))
        assert torch.equal(even, torch.cat([torch.full((3, 2), float(r)) for r in range(world)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_gather(total):
    mp.spawn(_worker, args=(2, _free_port(), total), nprocs=2, join=True)


def _eval_worker(rank, world, port, sizes):
    """The harness's cross-rank concatenation (test_last.evaluate): per class, every rank holds the results of its
    shard; gather_predictions must rebuild dataset order on every rank, including classes with fewer images than
    ranks."""
    import numpy as np
    from aaclip_hip.shard import gather_predictions
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for total in sizes:
            b, e = shard_range(total, rank, world)
            idx = np.arange(b, e)
            local = (np.stack([np.full((1, 6, 6), i, np.float32) for i in idx]) if len(idx) else np.zeros((0, 1, 6, 6), np.float32),
                     idx.astype(np.int64) % 2,
                     np.stack([np.full((6, 6), 0.5 * i, np.float32) for i in idx]) if len(idx) else np.zeros((0, 6, 6), np.float32),
                     (idx * 0.25).astype(np.float32))
            masks, labels, preds, scores = gather_predictions(local, total)
            ref = np.arange(total)
            assert masks.shape == (total, 1, 6, 6) and np.array_equal(masks[:, 0, 0, 0], ref.astype(np.float32))
            assert np.array_equal(labels, ref % 2) and labels.dtype == np.int64
            assert np.array_equal(preds[:, 3, 3], 0.5 * ref.astype(np.float32))
            assert np.array_equal(scores, (ref * 0.25).astype(np.float32))
    finally:
        dist.destroy_process_group()


def test_harness_gathers_sharded_predictions_in_dataset_order():
    mp.spawn(_eval_worker, args=(2, _free_port(), (7, 8, 1, 3)), nprocs=2, join=True)


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without torchrun: the process becomes a launcher, starts 2 child ranks before
    touching any GPU, and rank 0 reports n_gpus = ranks_seen = 2 with one all-gather of 2 x batch rows per step.
    CPU rehearsal of the control flow (gloo, stand-in step): the real path needs an MI355X."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "8", "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["global_batch"] == 16
    assert d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and d["steps"] == 3
    assert len(d["per_rank_images_per_s"]) == 2 and d["rows_all_gathered_per_step"] == 16
    assert d["launch"] == "self-launched child ranks"
    assert d["value"] is None and "REHEARSAL" in d["metric"]      # never mistaken for a measurement
    # one rank: no launcher, same line shape
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
                         "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["ranks_seen"] == 1 and d1["launch"] == "single process"


def test_bench_launcher_propagates_a_failing_rank():
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--precision", "nope", "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_rccl_selftest_launches_n_ranks_on_gloo():
    """tools/rccl_selftest.py --ranks N: the same self-launcher as bench.py; on CPU the gloo backend runs the same
    collectives (rank-distinct all-gather contents, fp64 times, ragged gather, barriers)."""
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "rccl_selftest.py"), "--ranks", "3", "--backend",
                        "gloo"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "world 3" in r.stdout and r.stdout.count("self-test ok") == 1


def test_bench_rehearsal_of_the_full_workload_two_ranks():
    """--workload full through the launcher (2 ranks, gloo rehearsal): the step ends in the all-gather of the image
    scores like the real full path; each rank caps its intra-op threads at cores / world."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "OMP_NUM_THREADS")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--workload", "full", "--rehearse-cpu"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["rows_all_gathered_per_step"] == 8
    assert "full AA-CLIP" in d["config"]["workload"] and d["value"] is None
    assert d["threads_per_rank"] == max(1, (os.cpu_count() or 1) // 2)


def test_pinned_feed_hands_out_alternating_batches_in_order():
    """bench.PinnedFeed on CPU (the rehearsal form: same call order, plain copies): batch i is host batch i & 1, a
    buffer is only refilled after done() of the step that used it, and two copies are always in flight ahead."""
    import sys
    import torch
    from conftest import REPO
    sys.path.insert(0, REPO)
    import bench
    feed = bench.PinnedFeed(torch, torch.device("cpu"), (3, 5), seed=1, rehearse=True)
    assert feed.issued == 2 and feed.taken == 0
    for i in range(7):
        x = feed.next()
        assert torch.equal(x, feed.host[i & 1]) and x is feed.dbuf[i & 1]
        assert feed.issued == i + 2                   # the refill of THIS buffer has not started yet
        feed.done()
        assert feed.issued == i + 3 and feed.taken == i + 1
    assert not torch.equal(feed.host[0], feed.host[1])


def test_bench_rehearsal_with_the_pinned_feed_two_ranks():
    """--feed pinned through the launcher (2 ranks, gloo rehearsal): every step takes its batch from the feed."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "8", "--feed", "pinned", "--rehearse-cpu"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["rows_all_gathered_per_step"] == 16 and "pinned host" in d["config"]["input"]
