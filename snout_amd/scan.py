"""Scan handlers: counterparts of snout/util/btle.py:22-122 (BtleScan) and the passive path of
snout/util/zigbee.py:26-70,194-209 + snout/core/radio.py:399-443 (ZigbeeScan / Scan.run), with the
external receivers (the ``btle_rx`` child, the GNU Radio flowgraph) replaced by libsnout_rx.so.

The scan loop semantics are the reference's: channels are visited in order; each packet goes
through ``handle_packet`` (which applies the same accept rule as BtleMessage.fromraw, appends to
``packets``, writes the dump file and emits ``btle.packet-received``); ``check_stop`` ends a
channel when ``packet_threshold`` packets were seen or ``timeout`` seconds elapsed.  With a
recorded/synthetic capture "elapsed" is capture time (sample_index / fs); with a live stream
(``StreamSource``: stdin, a FIFO) it is the wall clock, as in the reference.
"""
from __future__ import annotations

import os
import socket
import time
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np

from . import _ffi
from .message import BtleMessage, ZigbeeMessage
from .rx import SnoutRx, btle_format_line, rftap_encap

BTLE_FS = 4e6
ZIGBEE_FS = 4e6
SEGMENT = 1 << 24           # samples per processing segment
BTLE_OVERLAP = 1504         # longest BTLE packet, samples
ZIGBEE_OVERLAP = 17024 + 2048


class IqSource:
    """Where a scan gets its samples: ``read(channel)`` -> complex64 array for that channel."""

    def read(self, channel: int) -> np.ndarray:  # pragma: no cover - interface
        raise NotImplementedError


class FileSource(IqSource):
    """Capture file holding one channel: cf32 (GNU Radio file_sink format, sample_format 0), or
    interleaved int8 (1: `hackrf_transfer -r`, the input of upstream btle_rx) / int16 (2: USRP
    sc16) — integer captures go to the GPU as they are, read back as an [n, 2] array so that
    indexing is by sample."""

    def __init__(self, path: str, sample_format: int = 0):
        self.path = path
        self.sample_format = int(sample_format)

    def read(self, channel: int) -> np.ndarray:
        # memory-mapped: a scan touches one segment at a time, captures can be larger than RAM
        if os.path.getsize(self.path) == 0:
            return np.zeros(0, dtype=np.complex64)
        if self.sample_format == 0:
            return np.memmap(self.path, dtype=np.complex64, mode="r")
        a = np.memmap(self.path, dtype=np.int8 if self.sample_format == 1 else np.int16, mode="r")
        return a[:a.size // 2 * 2].reshape(-1, 2)


class StreamSource(IqSource):
    """A LIVE sample stream: stdin (``"-"``), a FIFO, a socket file -- anything that is read once, front to back, e.g.
    ``hackrf_transfer -r - | btle_rx -c 37 --format sc8 --iq -``.  This is how the reference's receivers get their samples
    (upstream btle_rx: a ring buffer of the radio's transfers processed half at a time with a tail of the longest packet,
    SURVEY Appendix A.1; the scan around it runs by the wall clock, snout/util/btle.py:111-122, snout/core/radio.py:399-443).

    ``segments()`` cuts the stream into segments of ``segment`` samples that overlap by the longest packet (802.15.4: plus the
    pre-roll of the DC filter, as the sharded scan does), uploads each as it completes and keeps two in flight on the GPU
    (submit / collect); it yields one record array per segment, in stream order, duplicates of the overlaps removed --
    the records of the same samples read from a file.  A scan over a StreamSource measures ``timeout`` by the wall clock
    (``live = True``), as the reference does; ``sample_index`` stays the position in the stream."""

    live = True

    def __init__(self, stream, sample_format: int = 0, segment: int = 1 << 22):
        self.sample_format = int(sample_format)
        self.segment = int(segment)
        if stream == "-":
            import sys
            stream = sys.stdin.buffer
        elif isinstance(stream, (str, bytes, os.PathLike)):
            stream = open(stream, "rb", buffering=0)
        self.stream = stream
        self.samples_read = 0

    def read(self, channel: int) -> np.ndarray:
        raise TypeError("a live stream is read once, in segments: use segments()")

    def _read_exact(self, nbytes: int) -> bytes:
        """Up to nbytes from the stream; shorter only at its end (a pipe hands over what the writer has flushed)."""
        parts, have = [], 0
        while have < nbytes:
            b = self.stream.read(nbytes - have)
            if not b:
                break
            parts.append(b)
            have += len(b)
        return b"".join(parts)

    def segments(self, proto: int, channel: int, device: int = -1, should_stop: Optional[Callable[[], bool]] = None, **rx_kw):
        import collections
        import torch
        bps = {0: 8, 1: 2, 2: 4}[self.sample_format]
        np_dt = {0: np.float32, 1: np.int8, 2: np.int16}[self.sample_format]
        zb = proto == _ffi.PROTO_ZIGBEE
        overlap = ZIGBEE_OVERLAP if zb else BTLE_OVERLAP
        preroll = 4 * 6250 if zb else 0            # sharded.ZIGBEE_PREROLL_CH: the DC estimate restarts with every segment
        seg = max(self.segment, 4 * (overlap + preroll))
        dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        rx = SnoutRx(proto=proto, channel=channel, device=device, sample_format=self.sample_format, **rx_kw)
        flight = collections.deque()                # (tensor kept alive, own_from)
        carry = b""
        first = 0                                   # stream position of the first sample of the next segment
        seen_until = -1
        recent = []

        def fresh(rec, own_from):
            """Drop what the segment before has reported (BTLE: by position; 802.15.4: same bytes within the tolerance
            the sharded scan uses) and what this segment found in its pre-roll."""
            nonlocal seen_until, recent
            keep = []
            now = []
            for i, p in enumerate(rec):
                si = int(p["sample_index"])
                if si < own_from:
                    continue
                if zb:
                    body = bytes(p["bytes"][:p["len"]])
                    if any(abs(si - s0) <= 8 * 64 + 8 and body == b0 for s0, b0 in recent):
                        continue
                    now.append((si, body))
                else:
                    if si <= seen_until:
                        continue
                    seen_until = si
                keep.append(i)
            if zb:
                recent = now
            return rec[keep]
        try:
            eof = False
            while not eof:
                data = self._read_exact(seg * bps - len(carry))
                buf = carry + data
                n = len(buf) // bps
                eof = len(data) < seg * bps - len(carry)
                self.samples_read += len(data) // bps
                if n and (len(data) or not first):
                    t = torch.frombuffer(bytearray(buf[:n * bps]), dtype={0: torch.float32, 1: torch.int8, 2: torch.int16}[self.sample_format]).to(dev)
                    rx.submit(t, first_sample_index=first)
                    flight.append((t, first + preroll if first else 0))
                    keep = overlap + preroll
                    if n > keep:
                        carry = buf[(n - keep) * bps:n * bps]
                        first += n - keep
                    else:
                        carry, eof = b"", True
                while len(flight) > (0 if eof else 1):
                    _, own = flight.popleft()
                    yield fresh(rx.collect(), own)
                    if should_stop is not None and should_stop():
                        return
        finally:
            rx.close()


class ArraySource(IqSource):
    def __init__(self, arrays: Dict[int, np.ndarray]):
        self.arrays = arrays

    def read(self, channel: int) -> np.ndarray:
        return self.arrays[channel]


class WidebandSource(IqSource):
    """One capture of the WHOLE band (BTLE: 80 Msps centred on 2442 MHz; 802.15.4: the 32 Msps
    synthetic raster, SURVEY §8d cfg #3 / #4): every channel is received in one pass through the
    polyphase channelizer -- what the reference does by hopping (snout/core/radio.py:415).  The
    capture is cut into overlapping segments that are pumped through the GPU with submit / collect
    (``ShardedScan``); with ``sharded=True`` under ``torch.distributed.run`` segment i goes to rank
    i mod N and rank 0 gets every record (cfg #5).  ``records(channel)`` then serves
    the scan loop channel by channel, in capture order."""

    def __init__(self, capture, proto: int, sample_format: int = 0, segment: int = 1 << 24,
                 sharded: bool = False, device: int = -1, batch: int = 4):
        self.capture = capture            # path of a capture file, or an array (complex64 / int pairs)
        self.proto = int(proto)
        self.sample_format = int(sample_format)
        self.segment = int(segment)
        self.sharded = bool(sharded)
        self.device = device
        self.batch = max(1, min(64, int(batch)))  # segments per submission (snout_rx_submit_batch_dev)
        self.rank = 0
        self._rec = None

    def _array(self) -> np.ndarray:
        if isinstance(self.capture, np.ndarray):
            return self.capture
        return FileSource(self.capture, self.sample_format).read(0)

    def _scan(self) -> np.ndarray:
        import torch
        import torch.distributed as tdist
        from .sharded import ShardedScan
        x = self._array()
        n_total = len(x)
        group_made = False
        if self.sharded and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not tdist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            backend = os.environ.get("SNOUT_BENCH_BACKEND", "nccl")
            torch.cuda.set_device(lr % max(1, torch.cuda.device_count()))
            tdist.init_process_group(backend)
            group_made = True
        self.rank = tdist.get_rank() if tdist.is_initialized() else 0
        dev = torch.device("cuda", torch.cuda.current_device())
        M = 40 if self.proto == _ffi.PROTO_BTLE else 16
        sc = ShardedScan(self.proto, n_channels=M, seg_len=self.segment, device=self.device,
                         batch=self.batch, sample_format=self.sample_format, **getattr(self, "rx_kw", {}))

        def source(a, b):
            chunk = np.ascontiguousarray(x[a:b])
            if chunk.dtype == np.complex64:
                chunk = chunk.view(np.float32)
            return torch.from_numpy(chunk.reshape(-1)).to(dev, non_blocking=False)
        try:
            rec = sc.run(n_total, source, gather_device=dev)
        finally:
            sc.close()
            if group_made:
                tdist.barrier()
                tdist.destroy_process_group()
        return rec if rec is not None else np.zeros(0, dtype=_ffi.PKT_DTYPE)

    def records(self, channel: int) -> np.ndarray:
        if self._rec is None:
            self._rec = self._scan()           # sorted by (proto, channel, sample_index), duplicates dropped
        r = self._rec
        return r[r["channel"] == channel]

    def read(self, channel: int) -> np.ndarray:
        raise TypeError("a wideband capture is not read per channel: use records(channel)")


def _pipelined_segments(proto: int, channel: int, x: np.ndarray, sample_format: int, device: int, **rx_kw):
    """One channel's capture through the pipelined submit / collect path (what bench.py measures):
    overlapping segments of SEGMENT samples, two in flight, uploaded chunk by chunk.  Yields the record
    array of every segment in capture order (duplicates of the overlaps still in)."""
    import torch
    from .sharded import ShardedScan
    dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
    sc = ShardedScan(proto, n_channels=1, channel=channel, seg_len=SEGMENT, device=device,
                     sample_format=sample_format, **rx_kw)

    def source(a, b):
        chunk = np.ascontiguousarray(x[a:b])
        if not chunk.flags.writeable:           # a read-only memmap: torch wants a buffer it may write to (it never does)
            chunk = chunk.copy()
        if chunk.dtype == np.complex64:
            chunk = chunk.view(np.float32)
        return torch.from_numpy(chunk.reshape(-1)).to(dev)
    try:
        sc.start(len(x), source)
        while sc.active():
            before = len(sc._parts)
            sc.step()
            for rec in sc._parts[before:]:
                yield rec
    finally:
        sc.close()


class _Events:
    """Minimal event bus (reference: snout/core/__init__.py:18-58)."""

    def __init__(self):
        self._handlers: Dict[str, List[Callable]] = {}

    def on(self, name: str, fn: Callable):
        self._handlers.setdefault(name, []).append(fn)

    def emit(self, name: str, **kw):
        for fn in self._handlers.get(name, []):
            fn(**kw)


class BtleScan:
    def __init__(self, channels: Iterable[int] = (37,), source: Optional[IqSource] = None,
                 timeout: Optional[float] = 10.0, packet_threshold: Optional[int] = None,
                 filename: Optional[str] = None, t0_epoch: Optional[float] = None,
                 access_addr: int = 0x8E89BED6, crc_init: int = 0x555555, device: int = -1):
        self.channels = list(channels)
        self.source = source
        self.timeout = timeout
        self.packet_threshold = packet_threshold
        self.filename = filename
        self.save_file = open(filename, "wb") if filename else None
        self.t0_epoch = time.time() if t0_epoch is None else t0_epoch
        self.access_addr = access_addr
        self.crc_init = crc_init
        self.device = device
        self.packets: List[BtleMessage] = []
        self.events = _Events()
        self._pkt_no = 0
        self._elapsed = 0.0
        self.start_time = None

    # -- the replaced child process: one receiver handle per channel -------------------------
    def lines(self, channel: int):
        """Yield btle_rx-grammar lines (bytes) for every PDU found on `channel`, CRC0 and CRC1
        alike, in capture order — what the reference reads from the child's stdout."""
        if hasattr(self.source, "records"):          # wideband capture: every channel was received at once
            for p in self.source.records(channel):
                self._elapsed = int(p["sample_index"]) / BTLE_FS
                yield btle_format_line(p, BTLE_FS, self.t0_epoch, self._pkt_no, self.access_addr)
                self._pkt_no += 1
            return
        if hasattr(self.source, "segments"):         # a live stream: segments as they arrive, timeout by the wall clock
            t_start = time.time()
            for rec in self.source.segments(_ffi.PROTO_BTLE, channel, self.device, access_addr=self.access_addr,
                                            crc_init=self.crc_init):
                for p in rec:
                    self._elapsed = time.time() - t_start
                    yield btle_format_line(p, BTLE_FS, self.t0_epoch, self._pkt_no, self.access_addr)
                    self._pkt_no += 1
                self._elapsed = time.time() - t_start
                if self.timeout and self._elapsed >= self.timeout:
                    return
            return
        x = self.source.read(channel)
        seen_until = -1
        for rec in _pipelined_segments(_ffi.PROTO_BTLE, channel, x, getattr(self.source, "sample_format", 0),
                                       self.device, access_addr=self.access_addr, crc_init=self.crc_init):
            for p in rec:
                si = int(p["sample_index"])
                if si <= seen_until:             # found again in the overlap of the next segment
                    continue
                seen_until = si
                self._elapsed = si / BTLE_FS
                yield btle_format_line(p, BTLE_FS, self.t0_epoch, self._pkt_no, self.access_addr)
                self._pkt_no += 1

    def run(self):
        self.start_time = time.time()
        for ch in self.channels:
            self._elapsed = 0.0
            n_before = len(self.packets)
            for line in self.lines(ch):
                self.handle_packet(line)
                if self.check_stop(n_before):
                    break
        return self.conclude()

    def handle_packet(self, line: bytes) -> bool:
        message = BtleMessage.fromraw(line)
        if not message:
            return False
        if self.save_file:
            self.save_file.write(line)
        self.packets.append(message)
        self.events.emit("btle.packet-received", message=message)
        return True

    def check_stop(self, n_before: int = 0) -> bool:
        if self.packet_threshold and len(self.packets) >= self.packet_threshold:
            return True
        if self.timeout and self._elapsed >= self.timeout:
            return True
        return False

    def conclude(self):
        if self.save_file:
            self.save_file.close()
            self.save_file = None
        return self.packets


class ZigbeeScan:
    """Passive 802.15.4 scan. Each frame is turned into the datagram the reference's flowgraph
    sends to UDP 127.0.0.1:52002 (RFtap header + MPDU); ``udp=True`` really sends it so an
    unmodified scapy-radio ``GnuradioSocket`` sniff loop receives it."""

    def __init__(self, channels: Iterable[int] = (11,), source: Optional[IqSource] = None,
                 timeout: Optional[float] = 10.0, packet_threshold: Optional[int] = None,
                 udp: bool = False, udp_addr=("127.0.0.1", 52002), t0_epoch: Optional[float] = None,
                 device: int = -1):
        self.channels = list(channels)
        self.source = source
        self.timeout = timeout
        self.packet_threshold = packet_threshold
        self.udp_addr = udp_addr
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM) if udp else None
        self.t0_epoch = time.time() if t0_epoch is None else t0_epoch
        self.device = device
        self.packets: List[ZigbeeMessage] = []
        self.events = _Events()
        self._elapsed = 0.0

    def frames(self, channel: int):
        if hasattr(self.source, "records"):          # wideband capture: every channel was received at once
            for p in self.source.records(channel):
                self._elapsed = int(p["sample_index"]) / ZIGBEE_FS
                yield p
            return
        if hasattr(self.source, "segments"):         # a live stream: segments as they arrive, timeout by the wall clock
            t_start = time.time()
            for rec in self.source.segments(_ffi.PROTO_ZIGBEE, channel, self.device, **getattr(self.source, "rx_kw", {})):
                for p in rec:
                    self._elapsed = time.time() - t_start
                    yield p
                self._elapsed = time.time() - t_start
                if self.timeout and self._elapsed >= self.timeout:
                    return
            return
        x = self.source.read(channel)
        recent = []                     # (sample_index, bytes) of the frames of the segment before
        # segments after the first start four DC-filter time constants early and leave what they find
        # there to the segment before (ShardedScan: ZIGBEE_PREROLL_CH, the same rule as the sharded scan)
        for rec in _pipelined_segments(_ffi.PROTO_ZIGBEE, channel, x, getattr(self.source, "sample_format", 0),
                                       self.device, **getattr(self.source, "rx_kw", {})):      # e.g. the lane shape (cli --lane-core)
            fresh = []
            for p in rec:
                si = int(p["sample_index"])
                body = bytes(p["bytes"][:p["len"]])
                # found again in the overlap: same bytes, the two runs' timing loops locked a few samples apart
                if any(abs(si - s0) <= 8 * 64 + 8 and body == b0 for s0, b0 in recent):
                    continue
                fresh.append((si, body))
                self._elapsed = si / ZIGBEE_FS
                yield p
            recent = fresh

    def run(self):
        for ch in self.channels:
            self._elapsed = 0.0
            for p in self.frames(ch):
                self.handle_packet(p)
                if self.check_stop():
                    break
        return self.conclude()

    def handle_packet(self, p) -> bool:
        dgram = rftap_encap(p)
        if self.sock:
            self.sock.sendto(dgram, self.udp_addr)
        m = ZigbeeMessage(channel=int(p["channel"]), mpdu=bytes(p["bytes"][:p["len"]]),
                          lqi=int(p["lqi"]), qual=int(p["lqi"]) / 255.0,
                          timestamp=self.t0_epoch + int(p["sample_index"]) / ZIGBEE_FS,
                          datagram=dgram)
        self.packets.append(m)
        self.events.emit("zigbee.packet-received", message=m)
        return True

    def check_stop(self) -> bool:
        if self.packet_threshold and len(self.packets) >= self.packet_threshold:
            return True
        if self.timeout and self._elapsed >= self.timeout:
            return True
        return False

    def conclude(self):
        if self.sock:
            self.sock.close()
            self.sock = None
        return self.packets
