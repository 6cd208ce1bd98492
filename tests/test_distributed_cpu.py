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
