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


def group_submissions(segs, batch: int, pad_to: int = 0):
    """Runs of up to ``batch`` consecutive segments of EQUAL length: what one
    ``snout_rx_submit_batch_dev`` call may carry.  Returns lists of indices into ``segs``; a segment of
    another length starts a new run.  ``pad_to``: a shorter segment counts as that long (the caller pads it with zeros)."""
    batch = max(1, int(batch))
    runs = []
    length = lambda j: max(segs[j][1] - segs[j][0], pad_to)
    for j, (a, b) in enumerate(segs):
        if runs and len(runs[-1]) < batch and length(j) == length(runs[-1][0]):
            runs[-1].append(j)
        else:
            runs.append([j])
    return runs


def shard_segments(n_total: int, seg_len: int, overlap: int, rank: int, world: int,
                   preroll: int = 0, uniform: bool = False) -> List[Tuple[int, int]]:
    """Cut [0, n_total) into segments of seg_len samples that each extend `overlap` samples into
    the next one (so a packet straddling a cut is whole in the earlier segment), and deal them
    round-robin: segment i -> rank i % world.  Returns this rank's [(start, stop), ...].
    ``preroll`` > 0 lets every segment but the first start that many samples early (a receiver
    with a start-up transient settles there; its records from before i * seg_len are the previous
    segment's to report).  ``uniform``: the first segment, which has nothing to pre-roll over, reads ``preroll`` samples
    more at its end instead, so that every segment that does not touch the capture's end has the same length (one batch)."""
    assert seg_len > 0 and overlap >= 0 and preroll >= 0 and 0 <= rank < world
    out = []
    i = 0
    start = 0
    while start < n_total:
        stop = min(start + seg_len + overlap + (preroll if (uniform and i == 0) else 0), n_total)
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


def wire_dtype(width: int) -> np.dtype:
    """Record layout on the wire: the first ``width`` bytes of a 160-byte ``snout_pkt``."""
    return np.dtype([("sample_index", "<u8"), ("proto", "<u4"), ("channel", "<u2"),
                     ("len", "<u2"), ("crc_ok", "u1"), ("lqi", "u1"), ("pdu_type", "u1"),
                     ("flags", "u1"), ("aux", "<u4"), ("bytes", "u1", (width - 24,))])


def widen_records(rec: np.ndarray) -> np.ndarray:
    """Wire records (any width) -> full 160-byte ``PKT_DTYPE`` records (zero padded)."""
    if rec.dtype == PKT_DTYPE:
        return rec
    out = np.zeros(rec.size, dtype=PKT_DTYPE)
    for f in ("sample_index", "proto", "channel", "len", "crc_ok", "lqi", "pdu_type", "flags", "aux"):
        out[f] = rec[f]
    out["bytes"][:, :rec["bytes"].shape[1]] = rec["bytes"]
    return out


_DROP = np.uint64(1) << np.uint64(62)      # sample_index of a record an append() disowned


def dedup_device(torch, rows, counts, cap: int, tol: int):
    """Fixed-shape sort + duplicate removal of gathered wire records on the device they sit on
    (no host synchronisation, so it can be queued on a side stream behind the all_gather).

    rows:   int64 [world * cap, width / 8]: the record slots of every rank, rank-major
    counts: int64 [world]: the valid prefix of every rank's block
    Returns (out int64 [world*cap + 1, width/8], n_keep int64 [1]): the kept records, sorted by
    (proto, channel, sample_index), in out[:n_keep].  Same rule as :func:`dedup_records`."""
    world = counts.numel()
    R = world * cap
    slot = torch.arange(cap, device=rows.device).repeat(world)
    valid = slot < counts.repeat_interleave(cap)
    si = rows[:, 0]
    valid &= si < int(_DROP)
    meta = rows[:, 1]                                           # proto | channel << 32 | len << 48
    key = ((meta & 0xF) << 60) | (((meta >> 32) & 0xFFFF) << 44) | (si & ((1 << 44) - 1))
    key = torch.where(valid, key, torch.full_like(key, (1 << 63) - 1))
    sk, order = torch.sort(key, stable=True)
    srt = rows.index_select(0, order)
    ok = sk != (1 << 63) - 1
    dup = torch.zeros(R, dtype=torch.bool, device=rows.device)
    if R > 1:
        near = ((sk[1:] >> 44) == (sk[:-1] >> 44)) & ((sk[1:] - sk[:-1]) <= tol) & ok[1:]
        if tol > 0:         # same frame only if length and bytes agree (dedup_records)
            near &= (srt[1:, 1] == srt[:-1, 1]) & (srt[1:, 3:] == srt[:-1, 3:]).all(dim=1)
        dup[1:] = near
    keep = ok & ~dup
    pos = torch.cumsum(keep, 0) - 1
    idx = torch.where(keep, pos, torch.full_like(pos, R))       # dropped rows land in the spare slot
    out = torch.empty((R + 1, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    out.index_copy_(0, idx, srt)
    return out, keep.sum().reshape(1)


class AsyncRecordGather:
    """Pipelined gather of packet records to rank 0.

    One exchange = ``begin()``, any number of ``append()`` (one per collected segment), ``launch()``;
    ``start(rec, dev_ptr)`` is the three in one.  ``launch()`` enqueues, on a side stream, the exchange
    (RCCL over xGMI with the nccl backend) and, on rank 0, an optional device-side sort + dedup of the
    gathered block (``dedup_tol`` not None) and its download into pinned host memory; ``finish()`` waits
    for the oldest launched exchange and returns the records (rank 0) or None.  Two exchanges may be in
    flight, so the exchange of step i overlaps the kernels of step i+1.

    The exchange is two collectives of fixed size: an ``all_gather`` of one 32-byte header per rank (true
    count, longest record) and a ``gather`` of ``cap`` record slots per rank TO RANK 0 -- only rank 0
    needs the records (round 3 sent every rank's block to every rank: 8 x the bytes into seven ranks that
    dropped them; ``SNOUT_GATHER=all`` keeps that form, one ``all_gather_into_tensor``, for comparison).
    A rank with more records than slots sends what fits and keeps the rest; since every rank sees every
    header, all ranks find out together in ``finish()``, exchange the remainders in a second
    (synchronous) all_gather and raise the capacity for the exchanges that follow -- overflow is a
    collective decision, never one rank raising while the others sit in the collective.

    ``width`` < 160 gathers only the first ``width`` bytes of each record (a BTLE record is at most 24 + 2 + 63 + 3
    bytes: 96 holds every record the decoder can emit, also a false access-address match with a 6-bit length on a
    data channel; the rest is zero by construction).  A record that does not fit is a ValueError -- like the
    overflow a collective one: its length travels in the header and every rank raises in ``finish()``.

    ``fake_world`` = F > 1 (world size 1 only; ``SNOUT_BENCH_FAKE_WORLD``): a rehearsal of rank 0's load at F ranks
    on one GPU -- after the real exchange rank 0's block is copied F - 1 more times (sample_index shifted by r 2^40 so
    that nothing de-duplicates away) and rank 0 sorts, de-duplicates and downloads F blocks, as it will at N = F.
    """

    HDR = 32        # header bytes per rank: count, longest record (host appends), longest record (device appends), launch ticket
    AHEAD = 4       # exchanges a rank may have opened behind the oldest unfinished one
    # process group -> [the group, gathers created on it so far (their creation index is part of the ticket), launches so far].
    # The entry HOLDS the group object (keyed by id() alone, a destroyed group's id could be reused by a new one,
    # whose gathers then started from stale counters); the default group's entry is dropped when torch.distributed has
    # been shut down or re-initialised since (its WORLD object changed).
    _groups = {}

    @classmethod
    def _registry(cls, group, dist):
        if group is None:
            world = dist.group.WORLD if dist.is_initialized() else None
            ent = cls._groups.get(0)
            if ent is None or ent[0] is not world:
                ent = cls._groups[0] = [world, 0, 0]
            return ent
        ent = cls._groups.get(id(group))
        if ent is None or ent[0] is not group:
            ent = cls._groups[id(group)] = [group, 0, 0]
        return ent

    def __init__(self, device=None, group=None, width: int = REC, dedup_tol: Optional[int] = None,
                 cap: int = 0, fake_world: int = 0, prealloc: int = 0):
        import os
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # with a process group the exchange is a real collective even at world size 1 (a one-GPU box then runs the
        # RCCL path that N > 1 runs); without one it is a device copy
        self.collective = dist.is_initialized()
        self.to_root = os.environ.get("SNOUT_GATHER", "root") != "all"
        self.root = dist.get_global_rank(group, 0) if (self.collective and group is not None) else 0
        self.on_gpu = self.backend == "nccl" or (self.backend == "none" and device is not None
                                                 and torch.device(device).type == "cuda")
        self.device = torch.device(device) if self.on_gpu else torch.device("cpu")
        self.width = int(width)
        assert 32 <= self.width <= REC and self.width % 8 == 0
        self.dedup_tol = dedup_tol
        self.fake = int(fake_world) if (fake_world and fake_world > 1 and self.world == 1) else 0
        self.blocks = self.fake or self.world       # record blocks rank 0 holds after an exchange
        self.cap = int(cap)             # capacity new exchanges are sized for (0: agreed at the first launch)
        # The capacity of an exchange is a function of its INDEX, not of when begin() happens to run: finish() of exchange j
        # may grow the capacity (every rank sees the same headers and computes the same number), and a rank may by then have
        # opened up to AHEAD exchanges behind j while another has not -- so the grown capacity holds from exchange
        # j + AHEAD + 1 on, on every rank alike (sized at call time, the send / receive sizes of one exchange
        # could differ between ranks: a hang).  begin() refuses to run further ahead than that.
        self.n_begun = self.n_finished = 0
        self.cap_from = []              # [(first exchange index, capacity)], ascending
        # exchanges of every gather that shares a process group must be launched in the same order on every rank: each launch
        # takes the group's next ticket, the ticket travels in the header and finish() compares it across the ranks
        ent = AsyncRecordGather._registry(group, dist)
        self.gid = ent[1]
        ent[1] += 1
        self.expect = None              # record counts of the last finished exchange: what the next download is sized for
        self.pool = []                  # finished slots, oldest first: reused once another has finished (views stay valid until then)
        self.cur = None                 # the open exchange (appends go here)
        self.closed = []                # complete, not yet launched
        self.inflight = []
        # high priority: the short exchange must not queue behind the next segment's kernels
        self.stream = torch.cuda.Stream(device=self.device, priority=-1) if self.on_gpu else None
        self.dtype = wire_dtype(self.width)
        # exchange buffers made up front (needs cap): one being filled, one complete, two in flight, one whose views the
        # consumer still reads -- pinned allocations of tens of MB do not belong into the first steps of a scan
        self.spare = [self._make_slot(self.cap) for _ in range(int(prealloc))] if (prealloc and self.cap) else []

    # ---- buffers -------------------------------------------------------------------------------
    def _make_slot(self, cap: int) -> dict:
        torch = self.torch
        W, pin, B = self.width, self.on_gpu, self.blocks
        head = torch.zeros(4, dtype=torch.int64, device=self.device)                 # this rank's header, on the device
        send = torch.zeros(cap * W, dtype=torch.uint8, device=self.device)
        heads = torch.zeros((B, 4), dtype=torch.int64, device=self.device)           # every rank's header
        recv = torch.zeros(B * cap * W, dtype=torch.uint8, device=self.device) if (self.rank == 0 or not self.to_root) else None
        host = n_host = None
        if self.rank == 0:
            rows = B * cap + 1 if self.dedup_tol is not None else B * cap
            host = torch.zeros(rows * W, dtype=torch.uint8, pin_memory=pin)
            n_host = torch.zeros(1, dtype=torch.int64, pin_memory=pin)
        ev = torch.cuda.Event() if self.on_gpu else None
        up = torch.cuda.Event() if self.on_gpu else None
        hdr2 = torch.zeros((B, 4), dtype=torch.int64, pin_memory=pin)                # every rank's header, downloaded
        hsend = torch.zeros(4, dtype=torch.int64, pin_memory=pin)                    # this rank's (count, longest on host, launch ticket), staged
        work = out = n_keep = None
        if self.rank == 0 and self.dedup_tol is not None and self._native():
            # the library's sort + duplicate removal (snout_records_dedup): its scratch, its output, its count
            work = torch.empty(self._lib.snout_records_dedup_workspace(B, cap), dtype=torch.uint8, device=self.device)
            out = torch.empty(B * cap * W, dtype=torch.uint8, device=self.device)
            n_keep = torch.zeros(1, dtype=torch.int64, device=self.device)
        return dict(work=work, out=out, n_keep=n_keep, cap=cap, send=send, head=head, heads=heads, recv=recv, hdr2=hdr2, hsend=hsend, host=host, n_host=n_host,
                    ev=ev, up=up, fill=0, n=0, rest=[], wide=0)

    def _native(self) -> bool:
        """The de-duplication runs in libsnout_rx.so (a handful of launches) when the exchange is on a GPU; the torch
        formulation of the same rule (:func:`dedup_device`, ~65 launches) serves CPU tensors (gloo tests)."""
        if not (self.on_gpu and self.width % 16 == 0):
            return False
        if getattr(self, "_lib", None) is None:
            from . import _ffi
            self._lib = _ffi.load()
        return True

    def _download(self, dst, src, nbytes: int) -> None:
        """Rank 0: `nbytes` of a device buffer into its pinned host mirror, on the exchange stream (inside _ctx).  The
        runtime's copy is a blit kernel that spreads over every free CU; a copy kernel of ours held to 4 / 8 / 16 workgroups
        (so that it fits the CUs `reserved_cus` leaves free) was tried and is slower: so few waves do not fill the PCIe
        link (step + 8-18 % against + 6 % with the blit in the 8-rank rehearsal, profiles/r4_fake_world.txt)."""
        if nbytes > 0:
            dst[:nbytes].copy_(src[:nbytes], non_blocking=True)

    def _agree(self, n: int) -> int:
        t = self.torch.tensor([n], dtype=self.torch.int64, device=self.device)
        if self.collective:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _ctx(self):
        return self.torch.cuda.stream(self.stream) if self.on_gpu else _null_ctx()

    @property
    def cap_scheduled(self) -> int:
        """The largest capacity agreed so far (in force now or from a later exchange on)."""
        return max([self.cap] + [c for _, c in self.cap_from])

    # ---- one exchange --------------------------------------------------------------------------
    def begin(self, n_hint: int = 0) -> None:
        """Open an exchange.  The first one agrees the capacity (collective: 1.25x the largest
        ``n_hint`` of any rank + 1024); later ones reuse or grow their slot without a collective."""
        assert self.cur is None, "close() or launch() the open exchange first"
        if self.cap == 0:
            m = self._agree(int(n_hint))
            self.cap = m + m // 4 + 1024
        if self.n_begun - self.n_finished > self.AHEAD:
            raise RuntimeError(f"{self.n_begun - self.n_finished} exchanges opened and not finished: finish() one first")
        k = self.n_begun
        self.n_begun += 1
        cap = self.cap                  # the capacity of exchange k: the last growth that has come into force
        for first, c in self.cap_from:
            if first <= k:
                cap = max(cap, c)
        self.cap_from = [(f, c) for f, c in self.cap_from if f > k]
        self.cap = cap
        slot = None
        if self.spare and self.spare[-1]["cap"] == cap:
            slot = self.spare.pop()
        elif len(self.pool) >= 2:         # the oldest finished slot: another exchange has finished since (its views were valid until then)
            slot = self.pool.pop(0)
            if slot["cap"] != cap:
                slot = None
        if slot is None:
            slot = self._make_slot(cap)
        slot["fill"], slot["n"], slot["rest"], slot["wide"], slot["index"] = 0, 0, [], 0, k
        with self._ctx():
            slot["head"].zero_()            # [2]: raised by the pack kernels of device-side appends
        self.cur = slot

    def append(self, rec, dev_ptr: int = 0, own_from: int = 0, rx=None) -> None:
        """Add the records of one collected segment.  ``rx``: the receiver handle that just collected
        them -- on a GPU its library then packs them from the device copy into the exchange buffer with
        ONE kernel launch (``snout_rx_pack_last_records``); ``rec`` may then be the record COUNT alone (a
        handle created with ``records_on_device=True``: nothing was downloaded).  ``dev_ptr``
        (SnoutRx.last_records_device()) does the same with torch operations.  Records with sample_index <
        ``own_from`` belong to another segment and are dropped."""
        torch = self.torch
        slot = self.cur
        counted = not isinstance(rec, np.ndarray)           # a count: the records are on the device only
        n = int(rec) if counted else int(rec.size)
        if n == 0:
            return
        if counted and not (self.on_gpu and rx is not None):
            raise TypeError("a record count needs the handle that holds the records (rx=) and a GPU exchange")
        # a record longer than the wire format holds: NOT an exception here (the other ranks would sit in the
        # collective): its length travels in the header and every rank raises together in finish()
        if not counted:
            slot["wide"] = max(slot["wide"], int(rec["len"].max()))
        W = self.width
        slot["n"] += n
        take = min(n, slot["cap"] - slot["fill"])
        if take < n:                                    # keeps its true count in the header; finish() resends
            if counted:                                 # rare: fetch what did not fit from the device copy
                ptr, have = rx.last_records_device()
                assert have == n
                raw = _device_bytes(torch, ptr + take * REC, (n - take) * REC, self.device).cpu().numpy()
                r = raw.view(PKT_DTYPE).copy()
                slot["wide"] = max(slot["wide"], int(r["len"].max()))
            else:
                r = np.ascontiguousarray(rec[take:]).copy()
            if own_from:
                r = r[r["sample_index"] >= own_from]
                slot["n"] -= (n - take) - r.size
            slot["rest"].append(r)
        if take == 0:
            return
        if self.on_gpu and rx is not None:
            dst = slot["send"].data_ptr() + slot["fill"] * W
            rx.pack_last_records(dst, take, W, int(own_from), self.stream.cuda_stream,
                                 longest_ptr=(slot["head"].data_ptr() + 16) if counted else 0)
            slot["fill"] += take
            return
        with self._ctx():
            view = slot["send"][slot["fill"] * W:(slot["fill"] + take) * W].view(take, W)
            if self.on_gpu and dev_ptr:
                view.copy_(_device_bytes(torch, dev_ptr, take * REC, self.device).view(take, REC)[:, :W])
            else:
                raw = np.ascontiguousarray(rec[:take]).view(np.uint8).reshape(take, REC)
                if self.on_gpu:     # contiguous upload, then the narrowing copy as a device kernel
                    view.copy_(torch.from_numpy(raw.copy()).to(self.device, non_blocking=True)[:, :W])
                else:
                    view.copy_(torch.from_numpy(raw[:, :W].copy()))
            if own_from:
                si = view.view(torch.int64)[:, 0]       # disowned records sort last and are not counted
                si.copy_(torch.where(si < own_from, torch.full_like(si, int(_DROP)), si))
        slot["fill"] += take

    def close(self) -> None:
        """The open exchange is complete (no collective is issued: ranks may close at different times)."""
        slot = self.cur
        self.cur = None
        with self._ctx():
            # header: true count and longest record appended on the host.  Staged in pinned memory (an asynchronous copy);
            # the slot is reused only after finish() has waited for this exchange's event.
            slot["hsend"][0], slot["hsend"][1] = slot["n"], slot["wide"]
            slot["head"][:2].copy_(slot["hsend"][:2], non_blocking=True)
            if self.on_gpu:
                slot["up"].record(self.stream)       # the callers' record buffers may be reused after this
        self.closed.append(slot)

    def launch(self) -> None:
        """Issue the collectives of the oldest complete exchange (closing the open one first if none is waiting).
        Every rank has to launch its exchanges -- of every gather that shares the process group -- in the same order."""
        torch = self.torch
        if not self.closed:
            self.close()
        if len(self.inflight) >= 2:
            raise RuntimeError("two gathers in flight: finish() one first")
        slot = self.closed.pop(0)
        W, cap, B = self.width, slot["cap"], self.blocks
        ent = AsyncRecordGather._registry(self.group, self.dist)
        ticket = ent[2]
        ent[2] += 1
        with self._ctx():
            heads = slot["heads"]
            slot["hsend"][2] = (self.gid << 40) | ticket         # (its own staging word: close()'s copy may still be reading the others)
            slot["head"][3:4].copy_(slot["hsend"][2:3], non_blocking=True)
            if self.collective:
                self.dist.all_gather_into_tensor(heads[:self.world].view(-1), slot["head"], group=self.group)
                if self.to_root:
                    outs = list(slot["recv"].view(B, cap * W)[:self.world].unbind(0)) if self.rank == 0 else None
                    self.dist.gather(slot["send"], outs, dst=self.root, group=self.group)
                else:
                    self.dist.all_gather_into_tensor(slot["recv"][:self.world * cap * W], slot["send"], group=self.group)
            else:
                heads[0].copy_(slot["head"])
                slot["recv"][:cap * W].copy_(slot["send"])
            if self.fake and self.rank == 0:
                # rehearsal: rank 0 holds F blocks as it will at N = F (its own, shifted so that nothing de-duplicates away)
                blk = slot["recv"].view(B, cap, W)
                blk[1:].copy_(blk[0:1].expand(B - 1, cap, W))
                si = blk.view(torch.int64)[1:, :, 0]
                si.add_((torch.arange(1, B, device=self.device, dtype=torch.int64) << 40).view(B - 1, 1))
                heads[1:].copy_(heads[0:1].expand(B - 1, 4))
            counts = heads[:, 0]
            slot["hdr2"].copy_(heads, non_blocking=True)         # every rank learns every count and every longest record
            if self.rank == 0:
                if self.dedup_tol is not None and slot["work"] is not None:
                    from . import _ffi
                    import ctypes as C
                    _ffi.check(self._lib.snout_records_dedup(
                        C.c_void_p(slot["recv"].data_ptr()), W, B, cap, C.c_void_p(heads.data_ptr()), 4, int(self.dedup_tol),
                        C.c_void_p(slot["out"].data_ptr()), C.c_void_p(slot["n_keep"].data_ptr()),
                        C.c_void_p(slot["work"].data_ptr()), slot["work"].numel(), C.c_void_p(self.stream.cuda_stream)))
                    # the download is sized by what the exchange before delivered (+ 2 % + 256 records): the count of THIS one is
                    # known on the device only, and every byte of a blit download is CU time taken from the receive path.
                    # finish() fetches the rest in the rare case that more arrived.
                    rows = B * cap if self.expect is None else min(B * cap, int(self.expect * 1.02) + 256)
                    slot["rows_down"] = rows
                    self._download(slot["host"], slot["out"], rows * W)
                    slot["n_host"].copy_(slot["n_keep"], non_blocking=True)
                elif self.dedup_tol is not None:
                    rows = slot["recv"].view(B * cap, W).view(torch.int64)
                    out, n_keep = dedup_device(torch, rows, torch.clamp(counts, max=cap), cap, int(self.dedup_tol))
                    slot["host"].copy_(out.view(torch.uint8).reshape(-1), non_blocking=True)
                    slot["n_host"].copy_(n_keep, non_blocking=True)
                else:
                    per = cap if self.expect is None else min(cap, int(self.expect * 1.02) + 256)       # rows per rank block
                    slot["rows_down"] = per
                    if per == cap:
                        self._download(slot["host"], slot["recv"], B * cap * W)
                    else:               # the first `per` rows of every block, one contiguous copy each (a strided device -> host
                        #                 copy goes through a staging buffer and blocks the host: 35 ms per 57 MB)
                        hv, rv = slot["host"].view(B, cap * W), slot["recv"].view(B, cap * W)
                        for r in range(B):
                            self._download(hv[r], rv[r], per * W)
            if self.on_gpu:
                slot["ev"].record(self.stream)
        self.inflight.append(slot)

    def start(self, rec: np.ndarray, dev_ptr: int = 0) -> None:
        """One segment = one exchange: begin + append + launch."""
        self.begin(int(rec.size))
        self.append(rec, dev_ptr)
        self.launch()

    def sync_uploads(self) -> None:
        """Block until the record buffers handed to append() have been read (they may be views of
        a receiver's pinned result slot that the next submit will overwrite)."""
        if self.on_gpu:
            for slot in self.closed + self.inflight:
                slot["up"].synchronize()

    def _exchange_rest(self, slot, counts) -> Optional[np.ndarray]:
        """Second, synchronous all_gather of what did not fit (every rank enters it: they all saw the
        same headers).  Returns the remainder records on rank 0."""
        torch = self.torch
        W, cap = self.width, slot["cap"]
        over = int(max(counts)) - cap
        mine = np.concatenate(slot["rest"]) if slot["rest"] else np.zeros(0, dtype=PKT_DTYPE)
        send = torch.zeros(over * W, dtype=torch.uint8)
        if mine.size:
            raw = np.ascontiguousarray(mine).view(np.uint8).reshape(mine.size, REC)[:, :W]
            send[:mine.size * W] = torch.from_numpy(raw.copy().reshape(-1))
        send = send.to(self.device)
        recv = torch.zeros(self.world * over * W, dtype=torch.uint8, device=self.device)
        if self.collective:
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
        else:
            recv.copy_(send)
        # same number on every rank; in force from the first exchange no rank can have opened yet (see __init__)
        self.cap_from.append((slot["index"] + self.AHEAD + 1, int(max(counts)) + int(max(counts)) // 4 + 1024))
        if self.rank != 0:
            return None
        host = recv.cpu().numpy().reshape(self.world, over * W)
        parts = [host[r, :max(0, int(counts[r]) - cap) * W].view(self.dtype) for r in range(self.world)]
        return np.concatenate(parts)

    def finish(self, views: bool = False):
        """Records of the oldest launched exchange (rank 0; None elsewhere), in the wire dtype
        (:func:`widen_records` pads them back to ``PKT_DTYPE``).  With ``dedup_tol`` set: one sorted,
        duplicate-free array.  Otherwise ``views=True`` returns a list with one zero-copy view per
        rank into the pinned receive buffer (valid until the next finish())
        instead of one concatenated array -- at 8 ranks the concatenation is a 30 MB host copy."""
        if not self.inflight:
            return None
        slot = self.inflight.pop(0)
        self.pool.append(slot)
        self.n_finished += 1
        if self.on_gpu:
            slot["ev"].synchronize()
        W, cap, B = self.width, slot["cap"], self.blocks
        h2 = slot["hdr2"].numpy()
        counts = [int(c) for c in h2[:self.world, 0]]
        tickets = {int(t) for t in h2[:self.world, 3]}
        if len(tickets) != 1:           # every rank sees the same headers: every rank raises
            raise RuntimeError("the gathers of this process group were launched in different orders on different ranks "
                               f"(tickets {sorted(tickets)}): launch every gather's exchanges in the same order everywhere")
        widest = int(h2[:self.world, 1:3].max())
        if widest > W - 24:             # every rank sees the same headers: every rank raises, none is left in a collective
            raise ValueError(f"a rank appended a record of {widest} bytes: it does not fit the {W}-byte wire format")
        rest = self._exchange_rest(slot, counts) if max(counts) > cap else None
        if self.rank != 0:
            return None
        if self.dedup_tol is not None:
            n = int(slot["n_host"].item())
            down = slot.get("rows_down")
            if down is not None and n > down:           # more than the download was sized for: fetch the rest now
                slot["host"][down * W:n * W].copy_(slot["out"][down * W:n * W])
            self.expect = n
            out = slot["host"].numpy()[:n * W].view(self.dtype)
            if rest is not None:        # rare: merge the remainder with the host rule
                out = dedup_records(np.concatenate([out, rest]), tol=int(self.dedup_tol))
            return [out] if views else out.copy()     # a view is valid until the next finish()
        allc = [int(c) for c in h2[:, 0]]
        down = slot.get("rows_down")
        most = min(max(allc), cap)
        if down is not None and most > down:            # more than the download was sized for: fetch the rest now
            for r in range(B):
                slot["host"].view(B, cap * W)[r, down * W:most * W].copy_(slot["recv"].view(B, cap * W)[r, down * W:most * W])
        self.expect = most
        host = slot["host"].numpy().reshape(B, cap * W)
        parts = [host[r, :min(allc[r], cap) * W].view(self.dtype) for r in range(B)]
        if rest is not None:
            parts.append(rest)
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
