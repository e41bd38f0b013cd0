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


def shard_segments(n_total: int, seg_len: int, overlap: int, rank: int, world: int,
                   preroll: int = 0) -> List[Tuple[int, int]]:
    """Cut [0, n_total) into segments of seg_len samples that each extend `overlap` samples into
    the next one (so a packet straddling a cut is whole in the earlier segment), and deal them
    round-robin: segment i -> rank i % world.  Returns this rank's [(start, stop), ...].
    ``preroll`` > 0 lets every segment but the first start that many samples early (a receiver
    with a start-up transient settles there; its records from before i * seg_len are the previous
    segment's to report)."""
    assert seg_len > 0 and overlap >= 0 and preroll >= 0 and 0 <= rank < world
    out = []
    i = 0
    start = 0
    while start < n_total:
        stop = min(start + seg_len + overlap, n_total)
        if i % world == rank:
            out.append((max(0, start - preroll), stop))
        start += seg_len
        i += 1
    return out


def dedup_records(rec: np.ndarray, tol: int = 0) -> np.ndarray:
    """Sort by (proto, channel, sample_index) and drop duplicates found by overlapping segments.
    ``tol`` > 0: records of the same channel with equal bytes whose sample_index differs by at most
    ``tol`` are the same frame (802.15.4: each run may first recognise a different one of the 8
    preamble symbols, 64 samples apart, and its timing loop locks at its own phase)."""
    if rec.size == 0:
        return rec
    si = rec["sample_index"]
    if int(si.max()) < (1 << 48):
        # one stable sort of a packed 64-bit key instead of a three-key lexsort
        key = (rec["proto"].astype(np.uint64) << np.uint64(60)) | (rec["channel"].astype(np.uint64) << np.uint64(48)) | si
        order = np.argsort(key, kind="stable")
        key = key[order]
        dup = ((key[1:] >> np.uint64(48)) == (key[:-1] >> np.uint64(48))) & (key[1:] - key[:-1] <= np.uint64(tol))
    else:
        order = np.lexsort((si, rec["channel"], rec["proto"]))
        p, c, x = rec["proto"][order], rec["channel"][order], si[order]
        dup = (p[1:] == p[:-1]) & (c[1:] == c[:-1]) & (x[1:] - x[:-1] <= np.uint64(tol))
    keep = np.ones(rec.size, dtype=bool)
    if tol > 0:
        # only candidate pairs pay for the comparison of their bytes
        idx = np.nonzero(dup)[0]
        a, b = rec[order[idx + 1]], rec[order[idx]]
        same = (a["len"] == b["len"]) & np.all(a["bytes"] == b["bytes"], axis=1)
        keep[idx[same] + 1] = False
    else:
        keep[1:] = ~dup
    return rec[order[keep]]


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


class AsyncRecordGather:
    """Pipelined gather of packet records to rank 0.

    ``start(rec)`` enqueues, on a side stream, the upload of this rank's records, ONE fixed-size
    ``all_gather`` (RCCL over xGMI with the nccl backend) and, on rank 0, the download of the
    gathered block into pinned host memory; ``finish()`` waits for the oldest started gather and
    returns the records (rank 0) or None.  Two gathers may be in flight, so the exchange of step i
    overlaps the kernels of step i+1.  Every rank sends ``cap`` record slots preceded by a header
    slot holding its count; ``cap`` is agreed once, at the first start(): 1.25x the largest count of
    any rank + 1024 (traffic of a capture is stationary; a rank that later exceeds it raises).  ``width`` < 160 gathers only the first ``width`` bytes of each record
    (BTLE records use at most 24 + 42 bytes; the rest is zero by construction).
    """

    def __init__(self, device=None, group=None, width: int = REC):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.on_gpu = self.backend == "nccl"
        self.device = device if self.on_gpu else torch.device("cpu")
        self.width = int(width)
        assert 24 < self.width <= REC and self.width % 8 == 0
        self.cap = 0
        self.slots = []
        self.inflight = []
        # high priority: the short exchange must not queue behind the next segment's kernels
        self.stream = torch.cuda.Stream(device=device, priority=-1) if self.on_gpu else None
        self.dtype = np.dtype([("sample_index", "<u8"), ("proto", "<u4"), ("channel", "<u2"),
                               ("len", "<u2"), ("crc_ok", "u1"), ("lqi", "u1"), ("pdu_type", "u1"),
                               ("flags", "u1"), ("aux", "<u4"), ("bytes", "u1", (self.width - 24,))])

    def _agree_cap(self, n: int):
        t = self.torch.tensor([n], dtype=self.torch.int64, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        self.cap = int(t.item()) + int(t.item()) // 4 + 1024
        torch = self.torch
        self.slots = []
        for _ in range(2):
            send = torch.zeros((self.cap + 1) * self.width, dtype=torch.uint8, device=self.device)
            # full 160-byte records land here first (one contiguous copy); the first `width` bytes
            # of each are then packed into `send` on the device
            stage = torch.zeros(self.cap * REC, dtype=torch.uint8, device=self.device) \
                if (self.on_gpu and self.width < REC) else None
            recv = torch.zeros(self.world * (self.cap + 1) * self.width, dtype=torch.uint8,
                               device=self.device)
            host = torch.zeros(recv.shape, dtype=torch.uint8,
                               pin_memory=self.on_gpu) if self.rank == 0 else None
            ev = torch.cuda.Event() if self.on_gpu else None
            up = torch.cuda.Event() if self.on_gpu else None
            self.slots.append(dict(send=send, recv=recv, host=host, ev=ev, up=up, stage=stage))
        self.next = 0

    def start(self, rec: np.ndarray, dev_ptr: int = 0) -> None:
        """``dev_ptr``: device address of the same records (SnoutRx.last_records_device()); with the
        nccl backend they are then packed straight from device memory, no upload."""
        torch = self.torch
        n = int(rec.size)
        while len(self.inflight) >= 2:
            raise RuntimeError("two gathers in flight: finish() one first")
        if not self.slots:
            self._agree_cap(n)      # collective: every rank makes its first start() together
        if n > self.cap:
            # re-agreeing is a collective every rank would have to enter at the same step
            raise RuntimeError(f"rank {self.rank}: {n} records exceed the agreed gather capacity "
                               f"{self.cap} (1.25x the largest first-step count + 1024)")
        slot = self.slots[self.next]
        self.next ^= 1
        hdr = np.zeros(self.width, dtype=np.uint8)
        hdr[:8] = np.frombuffer(np.uint64(n).tobytes(), dtype=np.uint8)
        raw = np.ascontiguousarray(rec).view(np.uint8).reshape(-1) if n else None    # n x 160 bytes, a view
        # No dependency on the caller's stream: the records are host memory that collect() has
        # already waited for, and the slot's buffers were released by the finish() of its last use.
        # (Waiting for the current stream would queue the exchange behind the front-end kernels of
        # the segments submitted since.)
        ctx = torch.cuda.stream(self.stream) if self.on_gpu else _null_ctx()
        with ctx:
            send = slot["send"]
            send[:self.width].copy_(torch.from_numpy(hdr), non_blocking=True)
            if n:
                view = send[self.width:(n + 1) * self.width].view(n, self.width)
                if self.on_gpu and dev_ptr:
                    dev = _device_bytes(torch, dev_ptr, n * REC, self.device)
                    view.copy_(dev.view(n, REC)[:, :self.width])
                elif slot["stage"] is not None:
                    # contiguous upload (a strided host-side gather of 52 k records costs ~8 ms), then
                    # the narrowing copy as a device kernel
                    st = slot["stage"][:n * REC]
                    st.copy_(torch.from_numpy(raw), non_blocking=True)
                    view.copy_(st.view(n, REC)[:, :self.width])
                else:
                    view.copy_(torch.from_numpy(raw.reshape(n, REC)[:, :self.width]), non_blocking=True)
            if self.on_gpu:
                slot["up"].record(self.stream)       # the caller's record buffer may be reused after this
            if self.world > 1:
                self.dist.all_gather_into_tensor(slot["recv"], send, group=self.group)
            else:
                slot["recv"].copy_(send)
            if self.rank == 0:
                slot["host"].copy_(slot["recv"], non_blocking=True)
            if self.on_gpu:
                slot["ev"].record(self.stream)
        self.inflight.append(slot)

    def sync_uploads(self) -> None:
        """Block until the record buffers handed to start() have been read (they may be views of
        a receiver's pinned result slot that the next submit will overwrite)."""
        if self.on_gpu:
            for slot in self.inflight:
                slot["up"].synchronize()

    def finish(self, views: bool = False):
        """Records of the oldest started gather (rank 0; None elsewhere).  ``views=True`` returns a
        list with one zero-copy view per rank into the pinned receive buffer (valid until the slot
        is reused two start() calls later) instead of one concatenated array -- at 8 ranks the
        concatenation is a 30 MB host copy per step."""
        if not self.inflight:
            return None
        slot = self.inflight.pop(0)
        if self.on_gpu:
            slot["ev"].synchronize()
        if self.rank != 0:
            return None
        host = slot["host"].numpy().reshape(self.world, (self.cap + 1) * self.width)
        parts = []
        for r in range(self.world):
            n = int(host[r, :8].view("<u8")[0])
            parts.append(host[r, self.width:(n + 1) * self.width].view(self.dtype))
        if views:
            return parts
        return np.concatenate(parts)


class _DevMem:
    """Minimal __cuda_array_interface__ holder so torch can view foreign device memory."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _device_bytes(torch, ptr: int, nbytes: int, device):
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
