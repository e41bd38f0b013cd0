"""Multi-GPU sharding of the receive path.

Units of work are capture segments (x channels); they are independent, so ranks never exchange
samples.  The only collective is the gather of decoded packet records (fixed 160-byte
``snout_pkt`` records) to rank 0: ``all_gather`` of the per-rank counts, then one padded
``all_gather`` of the records — RCCL over xGMI when the process group backend is ``nccl``,
``gloo`` on CPU (tests).  Payload is KB..MB per step, latency- not bandwidth-bound.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from ._ffi import PKT_DTYPE

REC = PKT_DTYPE.itemsize  # 160


def shard_segments(n_total: int, seg_len: int, overlap: int, rank: int, world: int
                   ) -> List[Tuple[int, int]]:
    """Cut [0, n_total) into segments of seg_len samples that each extend `overlap` samples into
    the next one (so a packet straddling a cut is whole in the earlier segment), and deal them
    round-robin: segment i -> rank i % world.  Returns this rank's [(start, stop), ...]."""
    assert seg_len > 0 and overlap >= 0 and 0 <= rank < world
    out = []
    i = 0
    start = 0
    while start < n_total:
        stop = min(start + seg_len + overlap, n_total)
        if i % world == rank:
            out.append((start, stop))
        start += seg_len
        i += 1
    return out


def dedup_records(rec: np.ndarray) -> np.ndarray:
    """Sort by (proto, channel, sample_index) and drop duplicates found by overlapping segments."""
    if rec.size == 0:
        return rec
    order = np.lexsort((rec["sample_index"], rec["channel"], rec["proto"]))
    rec = rec[order]
    key = np.stack([rec["proto"].astype(np.uint64), rec["channel"].astype(np.uint64),
                    rec["sample_index"]], axis=1)
    keep = np.ones(rec.size, dtype=bool)
    keep[1:] = np.any(key[1:] != key[:-1], axis=1)
    return rec[keep]


def gather_records(rec: np.ndarray, device=None, group=None) -> Optional[np.ndarray]:
    """Gather every rank's records on rank 0 (returns None on the other ranks).
    With the nccl backend the records travel GPU->GPU over xGMI."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = device if backend == "nccl" else torch.device("cpu")
    cnt = torch.tensor([rec.size], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts_h = counts.cpu().numpy()
    mx = int(counts_h.max())
    if mx == 0:
        return np.zeros(0, dtype=PKT_DTYPE) if rank == 0 else None
    buf = torch.zeros(mx * REC, dtype=torch.uint8, device=dev)
    if rec.size:
        src = torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1))
        buf[:rec.size * REC].copy_(src, non_blocking=True)
    allb = torch.empty(world * mx * REC, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allb, buf, group=group)
    if rank != 0:
        return None
    host = allb.cpu().numpy().reshape(world, mx * REC)
    parts = [host[r, :int(counts_h[r]) * REC].view(PKT_DTYPE) for r in range(world)]
    return np.concatenate(parts)
