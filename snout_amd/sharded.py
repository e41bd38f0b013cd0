"""Sharded scan of a long capture (SURVEY §8d cfg #5, §8e): the capture is cut into segments of
``seg_len`` input samples that overlap by the longest packet, segment i goes to rank i % world,
every rank runs its segments through its own GPU (pipelined submit/collect), the records are
gathered on rank 0 (RCCL with the nccl backend) and duplicates found in the overlaps are dropped.

``source(start, stop)`` returns the device tensor (interleaved float32, or complex64) holding input
samples [start, stop) — e.g. a slice of a resident capture, or an H2D upload of a file chunk.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import dist as sdist
from ._ffi import PKT_DTYPE, PROTO_BTLE
from .rx import SnoutRx

# longest packet in channel samples at 4 Msps + loop warm-up
BTLE_OVERLAP_CH = 1504
ZIGBEE_OVERLAP_CH = 17024 + 2048


class ShardedScan:
    def __init__(self, proto: int, n_channels: int = 1, channel: int = 37, seg_len: int = 1 << 24,
                 device: int = -1, **rx_kw):
        self.proto = proto
        self.n_channels = n_channels
        self.decim = n_channels // 2 if n_channels > 1 else 1
        self.pfb_taps = 16 * n_channels if n_channels > 1 else 0
        ov_ch = BTLE_OVERLAP_CH if proto == PROTO_BTLE else ZIGBEE_OVERLAP_CH
        self.overlap = ov_ch * self.decim + self.pfb_taps            # in input samples
        step = 2 * self.decim                                         # keep PFB phase parity aligned
        self.seg_len = max(step, seg_len // step * step)
        self.rx = SnoutRx(proto=proto, channel=channel, n_channels=n_channels, device=device, **rx_kw)

    def close(self):
        self.rx.close()

    def run(self, n_total: int, source: Callable[[int, int], "object"], group=None,
            gather_device=None) -> Optional[np.ndarray]:
        import torch.distributed as tdist
        world = tdist.get_world_size(group) if tdist.is_initialized() else 1
        rank = tdist.get_rank(group) if tdist.is_initialized() else 0
        segs = sdist.shard_segments(n_total, self.seg_len, self.overlap, rank, world)
        parts, pending = [], []
        for (a, b) in segs:
            x = source(a, b)
            pending.append(x)                                   # keep the tensor alive until collected
            self.rx.submit(x, first_sample_index=a // self.decim)
            if len(pending) == 2:
                parts.append(self.rx.collect())
                pending.pop(0)
        while pending:
            parts.append(self.rx.collect())
            pending.pop(0)
        mine = np.concatenate(parts) if parts else np.zeros(0, dtype=PKT_DTYPE)
        allrec = sdist.gather_records(mine, gather_device, group) if world > 1 else mine
        if allrec is None:
            return None
        return sdist.dedup_records(allrec, tol=0 if self.proto == PROTO_BTLE else 8 * 64 + 8)
