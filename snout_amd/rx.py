"""Receiver handle: the Python face of the C ABI (one handle = one configured receive path)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import PKT_DTYPE, PROTO_BTLE, PROTO_ZIGBEE, SnoutError  # noqa: F401


@dataclass
class Profile:
    ms_total: float
    ms_dominant: float
    dominant_launches: int
    n_hits: int
    bytes_algorithmic: int
    dominant_name: str


FMT_CF32, FMT_SC8, FMT_SC16 = 0, 1, 2
_NP_DTYPE = {FMT_CF32: np.float32, FMT_SC8: np.int8, FMT_SC16: np.int16}


class SnoutRx:
    """``SnoutRx(proto, channel=37)`` -> ``process(iq)`` returns a numpy record array
    (dtype :data:`PKT_DTYPE`). ``iq`` may be a numpy complex64/float32 array (host path,
    PCIe-inclusive) or a torch CUDA tensor / (device pointer, n) pair (HBM-resident path)."""

    def __init__(self, proto: int = PROTO_BTLE, channel: int = 37, n_channels: int = 1,
                 access_addr: int = 0, crc_init: int = 0, chip_threshold: int = 0,
                 taps_per_branch: int = 0, zb_core: int = 0, zb_warmup: int = 0,
                 max_hits: int = 0, device: int = -1, keep_channel_iq: bool = False,
                 sample_format: int = 0, batch_segments: int = 1, records_on_device: bool = False,
                 reserved_cus: int = 0):
        self._lib = _ffi.load()
        cfg = _ffi.RxCfg(abi_version=_ffi.ABI_VERSION, proto=proto, n_channels=n_channels,
                         taps_per_branch=taps_per_branch, channel=channel,
                         access_addr=access_addr, crc_init=crc_init,
                         chip_threshold=chip_threshold, zb_core=zb_core, zb_warmup=zb_warmup,
                         max_hits=max_hits, device=device)
        # SNOUT_CFG_KEEP_CHANNEL_IQ: unfused wideband kernels (CHAN_IQ tap); SNOUT_CFG_RECORDS_ON_DEVICE: collect() hands out
        # counts, the records stay on the device for pack_last_records() / last_records_device()
        cfg.flags = (_ffi.CFG_KEEP_CHANNEL_IQ if keep_channel_iq else 0) | (_ffi.CFG_RECORDS_ON_DEVICE if records_on_device else 0)
        self.records_on_device = bool(records_on_device)
        cfg.sample_format = int(sample_format)             # FMT_CF32 / FMT_SC8 / FMT_SC16
        cfg.batch_segments = int(batch_segments)           # segments one submit_batch() may carry
        cfg.reserved_cus = int(reserved_cus)               # CUs the channelizer's grid leaves to other streams
        self.batch_segments = max(1, int(batch_segments))
        self.sample_format = int(sample_format)
        self._h = C.c_void_p()
        _ffi.check(self._lib.snout_rx_create(C.byref(cfg), C.byref(self._h)))
        self.proto = proto
        self.n_channels = n_channels
        self._out = None
        self._out_ptr = None
        self._alloc_out(4096)

    def _alloc_out(self, cap: int):
        """Records land in page-locked memory owned by this handle (DMA target, no staging)."""
        if self._out_ptr:
            self._out = None
            self._lib.snout_host_free(self._out_ptr)
        self._out_ptr = self._lib.snout_host_alloc(cap * PKT_DTYPE.itemsize)
        if not self._out_ptr:
            raise MemoryError("snout_host_alloc failed")
        buf = (C.c_uint8 * (cap * PKT_DTYPE.itemsize)).from_address(self._out_ptr)
        self._out = np.frombuffer(buf, dtype=PKT_DTYPE)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.snout_rx_destroy(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_out_ptr", None):
            self._out = None
            self._lib.snout_host_free(self._out_ptr)
            self._out_ptr = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _run(self, fn, *args, cap: int, copy: bool = True):
        if self._out.size < cap:
            self._alloc_out(cap)
        while True:
            n_out = C.c_uint64(0)
            rc = fn(*args, C.c_void_p(self._out_ptr), self._out.size, C.byref(n_out))
            if rc == -5 and n_out.value > self._out.size:      # output capacity: grow and retry
                self._alloc_out(int(n_out.value) + 1024)
                continue
            _ffi.check(rc)
            view = self._out[:n_out.value]
            return view.copy() if copy else view

    def process(self, iq, first_sample_index: int = 0, stream: Optional[int] = None,
                copy: bool = True) -> np.ndarray:
        """Run one capture segment. ``copy=False`` returns a view of the handle's pinned record
        buffer, valid until the next call."""
        if isinstance(iq, np.ndarray):
            a = np.ascontiguousarray(iq)
            if a.dtype == np.complex64:
                a = a.view(np.float32)
            if a.dtype != _NP_DTYPE[self.sample_format]:
                raise TypeError(f"iq must be interleaved {_NP_DTYPE[self.sample_format].__name__} "
                                "(complex64 for cf32) for this handle's sample format")
            n = a.size // 2
            return self._run(lambda *r: self._lib.snout_rx_process(
                self._h, a.ctypes.data_as(C.c_void_p), n, first_sample_index, *r),
                cap=max(4096, n // 2048), copy=copy)
        # torch tensor on the GPU (complex64 [n] or float32 [2n])
        import torch
        if not (isinstance(iq, torch.Tensor) and iq.is_cuda and iq.is_contiguous()):
            raise TypeError("iq must be a numpy array or a contiguous torch CUDA tensor")
        n = self._tensor_samples(iq)
        st = stream if stream is not None else torch.cuda.current_stream(iq.device).cuda_stream
        return self._run(lambda *r: self._lib.snout_rx_process_dev(
            self._h, C.c_void_p(iq.data_ptr()), n, first_sample_index, C.c_void_p(st), *r),
            cap=max(4096, n // 2048), copy=copy)

    def _tensor_samples(self, iq) -> int:
        """Complex samples in a device tensor of this handle's sample format."""
        import torch
        want = {0: torch.float32, 1: torch.int8, 2: torch.int16}[self.sample_format]
        if self.sample_format == 0 and iq.dtype == torch.complex64:
            return iq.numel()
        if iq.dtype != want:
            raise TypeError(f"iq tensor must be interleaved {want} for this handle's sample format")
        return iq.numel() // 2

    # ---- pipelined form: up to three segments in flight ---------------------------------------
    def submit(self, iq, first_sample_index: int = 0, stream: Optional[int] = None) -> None:
        """Enqueue one device-resident segment (torch CUDA tensor) without waiting. The tensor must
        stay alive and unchanged until the matching :meth:`collect`."""
        import torch
        if not (isinstance(iq, torch.Tensor) and iq.is_cuda and iq.is_contiguous()):
            raise TypeError("submit() needs a contiguous torch CUDA tensor")
        n = self._tensor_samples(iq)
        st = stream if stream is not None else torch.cuda.current_stream(iq.device).cuda_stream
        _ffi.check(self._lib.snout_rx_submit_dev(self._h, C.c_void_p(iq.data_ptr()), n,
                                                 first_sample_index, C.c_void_p(st)))

    def submit_batch(self, iqs, first_sample_indices, min_sample_indices=None, stream: Optional[int] = None) -> None:
        """Enqueue several device-resident segments of EQUAL length as one submission (handle created
        with ``batch_segments`` >= len(iqs)); one :meth:`collect` returns the records of all of them,
        ordered by (segment, channel, sample_index).  ``min_sample_indices[k]``: records of segment k
        that start before it are dropped on the device."""
        import torch
        n = None
        for iq in iqs:
            if not (isinstance(iq, torch.Tensor) and iq.is_cuda and iq.is_contiguous()):
                raise TypeError("submit_batch() needs contiguous torch CUDA tensors")
            m = self._tensor_samples(iq)
            if n is not None and m != n:
                raise ValueError("the segments of a batch must have equal lengths")
            n = m
        k = len(iqs)
        if k == 0 or len(first_sample_indices) != k or (min_sample_indices is not None and len(min_sample_indices) != k):
            raise ValueError("one first_sample_index (and min_sample_index) per segment")
        st = stream if stream is not None else torch.cuda.current_stream(iqs[0].device).cuda_stream
        ptrs = (C.c_void_p * k)(*[iq.data_ptr() for iq in iqs])
        firsts = (C.c_uint64 * k)(*[int(v) for v in first_sample_indices])
        mins = (C.c_uint64 * k)(*[int(v) for v in min_sample_indices]) if min_sample_indices is not None else None
        _ffi.check(self._lib.snout_rx_submit_batch_dev(self._h, ptrs, k, n, firsts, mins, C.c_void_p(st)))

    def ready(self) -> bool:
        """True if the oldest submitted segment has finished: :meth:`collect` will not wait."""
        rc = self._lib.snout_rx_poll(self._h)
        if rc < 0:
            _ffi.check(rc)
        return rc == 1

    def collect(self, copy: bool = True):
        """Records of the oldest submitted segment. ``copy=False``: a view of the handle's pinned
        buffer, valid until three more submits.  A handle created with ``records_on_device=True`` returns the
        record COUNT (an int): the records stay in device memory."""
        ptr = C.c_void_p()
        n = C.c_uint64(0)
        _ffi.check(self._lib.snout_rx_collect_view(self._h, C.byref(ptr), C.byref(n)))
        if self.records_on_device:
            return int(n.value)          # the records stay on the device (pack_last_records / last_records_device)
        if n.value == 0:
            return np.zeros(0, dtype=PKT_DTYPE)
        buf = (C.c_uint8 * (n.value * PKT_DTYPE.itemsize)).from_address(ptr.value)
        view = np.frombuffer(buf, dtype=PKT_DTYPE)
        return view.copy() if copy else view

    def last_records_device(self):
        """(device pointer, count) of the records of the segment collected last — the device copy
        behind the array collect() returned; valid until three more submits."""
        ptr = C.c_void_p()
        n = C.c_uint64(0)
        _ffi.check(self._lib.snout_rx_last_records_dev(self._h, C.byref(ptr), C.byref(n)))
        return (ptr.value or 0), int(n.value)

    def pack_last_records(self, dst_ptr: int, dst_cap: int, width: int, own_from: int, stream: int,
                          skip: int = 0, longest_ptr: int = 0) -> int:
        """Pack the device copy of the last collected segment's records (from record ``skip`` on) into an exchange
        buffer on the device (wire format: the first ``width`` bytes of each record), on ``stream``; one launch.
        ``longest_ptr``: device uint64 raised to the largest record length packed."""
        n = C.c_uint64(0)
        _ffi.check(self._lib.snout_rx_pack_last_records(self._h, C.c_void_p(dst_ptr), dst_cap, width, own_from, skip,
                                                        C.c_void_p(longest_ptr or None), C.c_void_p(stream), C.byref(n)),
                   allow_overflow=True)
        return int(n.value)

    def soft(self, stage: int, channel_slot: int = 0, cap: int = 0) -> np.ndarray:
        cap = cap or (1 << 24)
        out = np.zeros(cap, dtype=np.float32)
        n = C.c_uint64(0)
        _ffi.check(self._lib.snout_rx_soft(self._h, stage, channel_slot,
                                           out.ctypes.data_as(C.c_void_p), cap, C.byref(n)),
                   allow_overflow=True)
        return out[:min(int(n.value), cap)]

    def profile(self) -> Profile:
        p = _ffi.RxProf()
        _ffi.check(self._lib.snout_rx_profile(self._h, C.byref(p)))
        return Profile(p.ms_total, p.ms_dominant, p.dominant_launches, p.n_hits,
                       p.bytes_algorithmic, p.dominant_name.decode())


    def profile_history(self) -> np.ndarray:
        """Dominant-kernel durations (ms) of the last <= 64 segments, oldest first."""
        ms = (C.c_float * 64)()
        n = C.c_uint32(0)
        _ffi.check(self._lib.snout_rx_profile_history(self._h, ms, 64, C.byref(n)))
        return np.array(ms[:n.value], dtype=np.float64)


def btle_format_line(pkt: np.void, fs_hz: float = 4e6, t0_epoch: float = 0.0, number: int = 0,
                     access_addr: int = 0x8E89BED6) -> bytes:
    lib = _ffi.load()
    rec = np.array([pkt], dtype=PKT_DTYPE)
    buf = C.create_string_buffer(512)
    n = lib.snout_btle_format_line(rec.ctypes.data_as(C.c_void_p), fs_hz, t0_epoch, number,
                                   access_addr, buf, 512)
    if n < 0:
        _ffi.check(n)
    return buf.raw[:n]


def rftap_encap(pkt: np.void) -> bytes:
    lib = _ffi.load()
    rec = np.array([pkt], dtype=PKT_DTYPE)
    buf = (C.c_uint8 * 256)()
    n = lib.snout_rftap_encap(rec.ctypes.data_as(C.c_void_p), buf, 256)
    if n < 0:
        _ffi.check(n)
    return bytes(buf[:n])
