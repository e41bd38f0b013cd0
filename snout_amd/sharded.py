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
# the single-pole DC estimate of the 802.15.4 chain (time constant 6250 samples) starts from zero in
# every segment, as it does once at the start of the reference's stream: segments after the first
# begin four time constants early and leave what they find there to the segment before
ZIGBEE_PREROLL_CH = 4 * 6250


class ShardedScan:
    """``handles`` > 1 keeps that many receiver handles, each on its own stream, and deals the
    segments to them in turn: short segments leave most of the GPU idle inside the latency-bound
    kernels (a 2^24-sample Zigbee segment is 128 waves of clock recovery), and independent handles
    let consecutive segments overlap.  ``batch`` > 1 (wideband handles) hands that many segments of
    equal length to the library as ONE submission (``snout_rx_submit_batch_dev``, up to 64): they run side by
    side inside the same kernels instead of one behind the other.  ``records_on_device=True`` (scans that
    feed a GPU ``sink``): the handles keep their records in device memory and ``collect`` hands out counts."""

    def __init__(self, proto: int, n_channels: int = 1, channel: int = 37, seg_len: int = 1 << 24,
                 device: int = -1, handles: int = 1, batch: int = 1, depth: int = 2, stream_priority: int = 0, **rx_kw):
        self.proto = proto
        self.n_channels = n_channels
        self.decim = n_channels // 2 if n_channels > 1 else 1
        self.pfb_taps = 16 * n_channels if n_channels > 1 else 0
        ov_ch = BTLE_OVERLAP_CH if proto == PROTO_BTLE else ZIGBEE_OVERLAP_CH
        self.overlap = ov_ch * self.decim + self.pfb_taps            # in input samples
        step = 2 * self.decim                                         # keep PFB phase parity aligned
        self.seg_len = max(step, seg_len // step * step)
        pre = 0 if proto == PROTO_BTLE else ZIGBEE_PREROLL_CH * self.decim
        self.preroll = (pre + step - 1) // step * step                # in input samples
        self.batch = max(1, int(batch))
        # Batched scans: every segment of a submission has to be equally long.  The first segment reads its pre-roll's
        # worth at its end instead (dist.shard_segments(uniform=True)), and a segment that the capture's end cuts short is
        # padded with zeros to the common length (it decodes nothing there): a capture's segments then go to the library as
        # ONE submission instead of three -- each submission costs a lane's whole serial clock recovery however few
        # samples it holds (cfg #5: three 802.15.4 submissions per step cost 3 x ~1.2 ms of zb_mm, profiles/r5_cfg5.md)
        # (done for every batch size, 1 included, so that what a capture decodes to does not depend on the batch -- but only
        #  for captures of MORE than one segment: a short capture is one segment of its own length, not a zero-padded 2^24)
        self.pad_to = self.seg_len + self.overlap + self.preroll
        self.stream_priority = int(stream_priority)                   # of the handles' streams (-1: high: the scan's small kernels go first)
        self.depth = max(1, min(3, int(depth)))                       # submissions in flight per handle (the library holds 3)
        if self.batch > 1:
            rx_kw = dict(rx_kw, batch_segments=self.batch)
        if n_channels > 1 and "reserved_cus" not in rx_kw:
            # One rank of several on the RCCL backend: the record gather's collectives complete only while the peers' kernels
            # run too, so they must not have to wait for a gap between two persistent channelizer launches that no other
            # rank shares -- eight CUs (one per RCCL channel) stay out of the channelizer's grid
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1 and tdist.get_backend() == "nccl":
                rx_kw = dict(rx_kw, reserved_cus=8)
        self.rxs = [SnoutRx(proto=proto, channel=channel, n_channels=n_channels, device=device, **rx_kw)
                    for _ in range(max(1, handles))]
        self.rx = self.rxs[0]
        self._streams = None
        self._segs = []

    def close(self):
        for rx in self.rxs:
            rx.close()

    @staticmethod
    def zero_pad(x, n_more: int):
        """``x`` (a device tensor of interleaved samples, any format) followed by ``n_more`` complex samples of zeros."""
        import torch
        return torch.cat([x.reshape(-1), torch.zeros(2 * n_more, dtype=x.dtype, device=x.device)])

    def my_segments(self, n_total: int, group=None):
        import torch.distributed as tdist
        world = tdist.get_world_size(group) if tdist.is_initialized() else 1
        rank = tdist.get_rank(group) if tdist.is_initialized() else 0
        return sdist.shard_segments(n_total, self.seg_len, self.overlap, rank, world, self.preroll, uniform=True)

    # ---- a scan as a sequence of steps, so that several scans can share a GPU (run_concurrent)
    def start(self, n_total: int, source: Callable[[int, int], "object"], group=None, sink=None,
              on_first=None, on_last=None) -> None:
        """Queue the submissions of one capture.  May be called again while submissions of the capture queued before
        are still in flight (a continuous scan: the GPU never drains between two captures); they are submitted and
        collected in order.

        ``sink``: an :class:`snout_amd.dist.AsyncRecordGather`; every collected segment's records are appended to it
        straight from device memory (no host concatenation) instead of being kept in ``_parts``.  ``on_first()`` is
        called before the first collected submission of THIS capture is appended (open the exchange there),
        ``on_last()`` after the last one (launch it)."""
        import collections
        import torch
        if self._streams is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            self._streams = [torch.cuda.Stream(device=dev, priority=self.stream_priority) for _ in self.rxs]
            self._jobs = collections.deque()        # submissions not yet submitted
            self._flight = collections.deque()      # submitted, not yet collected
            self._next = self._done = 0             # submissions submitted / collected since the handles were made
            self._appended = [None] * len(self.rxs)
            self._parts = []
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())               # the capture was produced on this stream
        for st in self._streams:
            st.wait_event(ready)
        import torch.distributed as tdist
        self._segs = self.my_segments(n_total, group)
        # a capture that fits one segment is submitted at its own length (padded to seg_len it cost a full
        # segment of channelizer and lane work, and a frame cut by its end was pushed on through the zeros)
        pad_to = self.pad_to if n_total > self.pad_to else 0
        world = tdist.get_world_size(group) if tdist.is_initialized() else 1
        rank = tdist.get_rank(group) if tdist.is_initialized() else 0
        # channel-sample index below which segment j's records belong to the segment before it
        own_from = [((rank + j * world) * self.seg_len) // self.decim if (rank + j * world) else 0
                    for j in range(len(self._segs))]
        if not self.active():
            self._parts = []
        # submissions: runs of up to `batch` consecutive segments of equal length
        subs = sdist.group_submissions(self._segs, self.batch, pad_to)
        for k, sub in enumerate(subs):
            self._jobs.append(dict(segs=[self._segs[i] for i in sub], own=[own_from[i] for i in sub], source=source, sink=sink, pad_to=pad_to,
                                   on_first=on_first if k == 0 else None, on_last=on_last if k == len(subs) - 1 else None))
        if not subs and (on_first is not None or on_last is not None):
            # nothing to do for this rank: the exchange still opens and closes -- in its turn, behind the submissions of
            # the capture before (an empty job carries the callbacks; called here they would open the next exchange while
            # the previous one is still being filled)
            self._jobs.append(dict(segs=[], own=[], source=source, sink=sink, pad_to=0, on_first=on_first, on_last=on_last))

    def active(self) -> bool:
        return self._streams is not None and bool(self._jobs or self._flight)

    def backlog(self) -> int:
        """Submissions queued or in flight."""
        return 0 if self._streams is None else len(self._jobs) + len(self._flight)

    def step(self, block: bool = True) -> bool:
        """Submit the next segment if a slot is free (``depth`` per handle), else collect the oldest.
        ``block=False``: collect only if it has finished; returns False if nothing could be done."""
        import torch
        import time
        H = len(self.rxs)
        n_fl = len(self._flight)
        tr = getattr(self, "trace", None)           # dev aid: a list that receives (what, seconds) per host action
        t0 = time.perf_counter() if tr is not None else 0.0
        if (not block and n_fl >= H and self.rxs[self._done % H].ready()):
            pass                                    # results are waiting: take them before queueing more
        elif self._jobs and not self._jobs[0]["segs"]:
            # an empty job (a capture without a segment for this rank): no handle, no slot; it takes its turn in the
            # collection order
            self._flight.append(self._jobs.popleft())
            return True
        elif self._jobs and n_fl < self.depth * H:
            j = self._next
            job = self._jobs.popleft()
            st = self._streams[j % H]
            if self._appended[j % H] is not None:
                # the result slot this submit reuses may still be being read by the sink's copy
                st.wait_event(self._appended[j % H])
                self._appended[j % H] = None
            with torch.cuda.stream(st):
                xs = [job["source"](a, b) for a, b in job["segs"]]
                if job["pad_to"]:
                    xs = [x if (b - a) >= job["pad_to"] else self.zero_pad(x, job["pad_to"] - (b - a)) for x, (a, b) in zip(xs, job["segs"])]
                if self.batch > 1:
                    # the library drops what a segment finds before its own range (its pre-roll)
                    self.rxs[j % H].submit_batch(xs, [a // self.decim for a, _ in job["segs"]],
                                                 [o if self.preroll else 0 for o in job["own"]],
                                                 stream=st.cuda_stream)
                else:
                    self.rxs[j % H].submit(xs[0], first_sample_index=job["segs"][0][0] // self.decim,
                                           stream=st.cuda_stream)
            job["alive"] = xs                        # the tensors stay alive until the submission is collected
            self._flight.append(job)
            self._next += 1
            if tr is not None:
                tr.append(("submit%d x%d" % (self.proto, len(xs)), time.perf_counter() - t0))
            return True
        if n_fl and not self._flight[0]["segs"]:
            job = self._flight.popleft()
            if job["on_first"] is not None:
                job["on_first"]()
            if job["on_last"] is not None:
                job["on_last"]()
            return True
        if n_fl:
            j = self._done
            rx = self.rxs[j % H]
            if not block and not rx.ready():
                return False
            job = self._flight.popleft()
            own = job["own"][0] if (self.preroll and self.batch == 1) else 0
            sink = job["sink"]
            if job["on_first"] is not None:
                job["on_first"]()
            if sink is not None:
                rec = rx.collect(copy=False)                    # segments of a handle complete in order
                sink.append(rec, own_from=own, rx=rx)
                if sink.on_gpu:
                    ev = torch.cuda.Event()
                    ev.record(sink.stream)
                    self._appended[j % H] = ev
            else:
                if getattr(rx, "records_on_device", False):
                    raise RuntimeError("this scan keeps its records on the device (records_on_device=True): start() it with a "
                                       "sink (an AsyncRecordGather on the GPU) -- run() / run_concurrent() collect to the host")
                rec = rx.collect()
                if own:
                    rec = rec[rec["sample_index"] >= own]
                self._parts.append(rec)
            job["alive"] = None
            self._done += 1
            if job["on_last"] is not None:
                job["on_last"]()
            if tr is not None:
                tr.append(("collect%d%s" % (self.proto, " (blocking)" if block else ""), time.perf_counter() - t0))
            return True
        return False

    def finish(self, parts, group=None, gather_device=None) -> Optional[np.ndarray]:
        """Gather this rank's records on rank 0 and drop the duplicates of the overlaps."""
        import torch.distributed as tdist
        world = tdist.get_world_size(group) if tdist.is_initialized() else 1
        mine = np.concatenate(parts) if parts else np.zeros(0, dtype=PKT_DTYPE)
        allrec = sdist.gather_records(mine, gather_device, group) if world > 1 else mine
        if allrec is None:
            return None
        return sdist.dedup_records(allrec, tol=0 if self.proto == PROTO_BTLE else 8 * 64 + 8)

    def run(self, n_total: int, source: Callable[[int, int], "object"], group=None,
            gather_device=None, stats: Optional[dict] = None) -> Optional[np.ndarray]:
        """``stats`` (optional dict) receives ``device_s``: wall time of the submit/collect loop, i.e.
        until this rank's last records are in host memory, and ``post_s``: gather + sort + dedup."""
        import time
        t0 = time.perf_counter()
        self.start(n_total, source, group)
        while self.active():
            self.step()
        t1 = time.perf_counter()
        out = self.finish(self._parts, group, gather_device)
        if stats is not None:
            stats["device_s"], stats["post_s"] = t1 - t0, time.perf_counter() - t1
        return out


def pump(scans) -> None:
    """Drive started scans to completion from one host thread: every scan submits while it has a free
    slot and collects what has finished; the thread waits (in one scan's collect) only when no scan can
    do either -- a blocking collect per turn kept the other scan's queue from being refilled."""
    while True:
        live = [sc for sc in scans if sc.active()]
        if not live:
            return
        if not any([sc.step(block=False) for sc in live]):
            live[0].step()


def run_concurrent(scans, n_totals, sources, group=None, gather_device=None, stats: Optional[dict] = None):
    """SURVEY §8d cfg #5: several wideband scans (e.g. BTLE 40 channels and Zigbee 16 channels, each
    with its own capture) share every GPU.  Every handle of every scan has its own stream and the
    scans take turns submitting / collecting segments, so the kernels of one fill the gaps of the
    other.  Returns one record array per scan (rank 0; None elsewhere)."""
    import time
    t0 = time.perf_counter()
    for sc, n, src in zip(scans, n_totals, sources):
        sc.start(n, src, group)
    pump(scans)
    t1 = time.perf_counter()
    out = [sc.finish(sc._parts, group, gather_device) for sc in scans]
    if stats is not None:
        stats["device_s"], stats["post_s"] = t1 - t0, time.perf_counter() - t1
    return out
