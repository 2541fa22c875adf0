"""Frame-parallel sharding across the GPUs of one node: one process per GPU, contiguous ranges of pairs per rank, no
data-path collective, and ONE all-gather of the fixed 32-byte per-pair result records at the end of a batch
(RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).  The reference has no counterpart (it is a
single-process loop); SURVEY.md section 8(e).

torch is imported only here and only when a process group is wanted: the single-GPU path never loads it.  When it is
used, import this module (i.e. torch) BEFORE mavflow._lib so that libmavflow binds to the HIP runtime torch loaded."""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np

RECORD_BYTES = 32


def shard(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `n_items` for `rank`; the first n_items % world ranks take one extra item."""
    if world < 1 or not (0 <= rank < world) or n_items < 0:
        raise ValueError(f"bad shard request n={n_items} rank={rank} world={world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def env_ranks() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str):
    """torch.distributed over env:// (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE set by torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    rank, world, local_rank = env_ranks()
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist, rank, world, local_rank


def allgather_records(dist, local, out=None):
    """All ranks' uint8 record tensors concatenated in rank order (every rank contributes the same count)."""
    import torch
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def allgather_numpy(dist, records: np.ndarray) -> np.ndarray:
    """CPU convenience (gloo): structured per-pair records in, all ranks' records out, rank order."""
    import torch
    local = torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).reshape(-1).copy())
    out = allgather_records(dist, local)
    return out.numpy().view(records.dtype).reshape(-1)


def allgather_pairs(dist, records: np.ndarray, n_items: int) -> np.ndarray:
    """All n_items per-pair records in PAIR order from every rank's shard (`records` = this rank's shard(n_items, rank, world)
    slice, possibly empty).  Ragged totals: every rank pads its shard to ceil(n_items / world) records so that the collective
    stays ONE fixed-size all-gather, and the padding is dropped on arrival using the same shard table."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard(n_items, rank, world)
    if len(records) != hi - lo:
        raise ValueError(f"rank {rank} holds {len(records)} records, its shard of {n_items} is [{lo}, {hi})")
    per = -(-n_items // world) if n_items else 0
    if per == 0:
        return records[:0].copy()
    padded = np.zeros(per, records.dtype)
    padded[:hi - lo] = records
    allrec = allgather_numpy(dist, padded).reshape(world, per)
    return np.concatenate([allrec[r, :shard(n_items, r, world)[1] - shard(n_items, r, world)[0]] for r in range(world)])
