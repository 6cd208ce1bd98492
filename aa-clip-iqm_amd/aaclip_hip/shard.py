"""Data-parallel sharding of the image batch: images are
independent, so the global batch is split contiguously by rank, weights are
replicated, and the only communication is one all-gather of per-rank results
(RCCL over xGMI on MI355X: backend "nccl"; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` images owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally sized per-rank results along dim 0 with ONE all-gather."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    return _all_gather(local, dist.get_world_size(group), group)


def _all_gather(local: torch.Tensor, world: int, group=None) -> torch.Tensor:
    import torch.distributed as dist
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def gather_ragged_rows(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """As gather_rows for shards from shard_range (sizes differ by at most one): pad to the
    largest shard, one all-gather, drop the padding."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    mx = (total + world - 1) // world
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    allr = gather_rows(pad, group)
    parts = []
    for r in range(world):
        b, e = shard_range(total, r, world)
        parts.append(allr[r * mx: r * mx + (e - b)])
    return torch.cat(parts, dim=0)


def gather_predictions(arrays, total: int, device=None, group=None):
    """Evaluation harness across ranks: every rank evaluated the images shard_range(total, rank, world)
    of one class; `arrays` is its tuple of per-image numpy arrays (masks, labels, anomaly maps, image scores -- any
    dtypes and trailing shapes, first dimension = its shard size).  Returns the tuple of full arrays in dataset order
    on every rank, using ONE all-gather per array (ragged shards are padded to the largest, see gather_ragged_rows).
    `device`: where the collective runs ("cuda:N" for RCCL, None/cpu for gloo)."""
    import numpy as np
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tuple(arrays)
    out = []
    for a in arrays:
        t = torch.from_numpy(np.ascontiguousarray(a))
        if device is not None:
            t = t.to(device)
        out.append(gather_ragged_rows(t, total, group).cpu().numpy())
    return tuple(out)
