"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Builds the library with the committed Makefile if it is missing or stale.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


class SnoutPkt(C.Structure):
    _fields_ = [("sample_index", C.c_uint64), ("proto", C.c_uint32), ("channel", C.c_uint16),
                ("len", C.c_uint16), ("crc_ok", C.c_uint8), ("lqi", C.c_uint8),
                ("pdu_type", C.c_uint8), ("flags", C.c_uint8), ("aux", C.c_uint32),
                ("bytes", C.c_uint8 * 136)]


assert C.sizeof(SnoutPkt) == 160

PKT_DTYPE = np.dtype([("sample_index", "<u8"), ("proto", "<u4"), ("channel", "<u2"),
                      ("len", "<u2"), ("crc_ok", "u1"), ("lqi", "u1"), ("pdu_type", "u1"),
                      ("flags", "u1"), ("aux", "<u4"), ("bytes", "u1", (136,))])
assert PKT_DTYPE.itemsize == 160


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE)
            if f.endswith((".c", ".h", ".inc")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "snout_rx.h"))
    stale = force or not os.path.exists(_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "liboracle.so"], check=True)
    return _LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        u8p, u64p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        _lib.oracle_btle_whiten_seq.argtypes = [C.c_uint32, u8p, C.c_int]
        _lib.oracle_btle_whiten_seq.restype = None
        _lib.oracle_btle_crc24.argtypes = [u8p, C.c_int, C.c_uint32]
        _lib.oracle_btle_crc24.restype = C.c_uint32
        _lib.oracle_btle_crc24_table.argtypes = [u8p, C.c_int, C.c_uint32]
        _lib.oracle_btle_crc24_table.restype = C.c_uint32
        _lib.oracle_btle_crc_bytes.argtypes = [C.c_uint32, u8p]
        _lib.oracle_btle_crc_bytes.restype = None
        _lib.oracle_btle_bits.argtypes = [f32p, C.c_uint64, u8p]
        _lib.oracle_btle_bits.restype = C.c_uint64
        _lib.oracle_btle_all_hits.argtypes = [u8p, C.c_uint64, C.c_uint32, u64p, C.c_uint64]
        _lib.oracle_btle_all_hits.restype = C.c_uint64
        _lib.oracle_btle_segment.argtypes = [f32p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_void_p, C.c_uint64, u64p,
                                             u64p, C.c_uint64, u64p]
        _lib.oracle_btle_segment.restype = C.c_int
        _lib.oracle_zigbee_segment.argtypes = [f32p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
        _lib.oracle_zigbee_segment.restype = C.c_int
        _lib.oracle_zigbee_lane_soft.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, f32p, f32p, C.c_uint64, u64p]
        _lib.oracle_zigbee_lane_soft.restype = C.c_int
        _lib.oracle_zb_discrim.argtypes = [f32p, C.c_uint64, f32p]
        _lib.oracle_zb_discrim.restype = None
        _lib.oracle_fast_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.oracle_fast_atan2f.restype = C.c_float
        _lib.oracle_zb_chip_map.restype = C.POINTER(C.c_uint32)
        _lib.oracle_zb_mmse_taps.restype = C.POINTER(C.c_float)
        _lib.oracle_crc16_154.argtypes = [u8p, C.c_int]
        _lib.oracle_crc16_154.restype = C.c_uint16
        _lib.oracle_pfb_nout.argtypes = [C.c_uint64, C.c_uint32]
        _lib.oracle_pfb_nout.restype = C.c_uint64
        _lib.oracle_pfb_proto.argtypes = [C.c_uint32]
        _lib.oracle_pfb_proto.restype = C.POINTER(C.c_float)
        _lib.oracle_pfb.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, C.c_uint64]
        _lib.oracle_pfb.restype = C.c_int
        _lib.oracle_pfb_block_order.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, C.c_uint64]
        _lib.oracle_pfb_block_order.restype = C.c_int
        _lib.oracle_pfb_legacy_fft.argtypes = [C.c_int]
        _lib.oracle_pfb_legacy_fft.restype = None
        _lib.oracle_btle_bin_channel.argtypes = [C.c_uint32]
        _lib.oracle_btle_bin_channel.restype = C.c_uint32
        _lib.oracle_zigbee_bin_channel.argtypes = [C.c_uint32]
        _lib.oracle_zigbee_bin_channel.restype = C.c_uint32
        _lib.oracle_wideband_segment.argtypes = [f32p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.c_uint64, u64p]
        _lib.oracle_wideband_segment.restype = C.c_int
        _lib.oracle_set_threads.argtypes = [C.c_int]
        _lib.oracle_set_threads.restype = C.c_int
        _lib.oracle_hw_threads.restype = C.c_int
        _lib.oracle_narrowband_parallel.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        _lib.oracle_narrowband_parallel.restype = C.c_int
        _lib.oracle_wideband_parallel.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        _lib.oracle_wideband_parallel.restype = C.c_int
    return _lib


def _f32(iq: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(iq)
    if a.dtype == np.complex64:
        a = a.view(np.float32)
    assert a.dtype == np.float32
    return a


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def btle_whiten_seq(channel: int, n: int = 42) -> bytes:
    out = np.zeros(n, dtype=np.uint8)
    lib().oracle_btle_whiten_seq(channel, _p(out, C.c_uint8), n)
    return out.tobytes()


def btle_crc24(data: bytes, init: int = 0x555555, table: bool = False) -> int:
    a = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    f = lib().oracle_btle_crc24_table if table else lib().oracle_btle_crc24
    return int(f(_p(a, C.c_uint8), a.size, init))


def btle_crc_bytes(r: int) -> bytes:
    out = np.zeros(3, dtype=np.uint8)
    lib().oracle_btle_crc_bytes(r, _p(out, C.c_uint8))
    return out.tobytes()


def btle_bits(iq: np.ndarray) -> np.ndarray:
    a = _f32(iq)
    n = a.size // 2
    bits = np.zeros(max(n, 1), dtype=np.uint8)
    nb = lib().oracle_btle_bits(_p(a, C.c_float), n, _p(bits, C.c_uint8))
    return bits[:nb]


def btle_all_hits(bits: np.ndarray, aa: int = 0x8E89BED6) -> np.ndarray:
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    cap = max(1024, bits.size // 64)
    hits = np.zeros(cap, dtype=np.uint64)
    n = lib().oracle_btle_all_hits(_p(bits, C.c_uint8), bits.size, aa, _p(hits, C.c_uint64), cap)
    assert n <= cap
    return hits[:n]


def btle_segment(iq: np.ndarray, channel: int = 37, aa: int = 0x8E89BED6, crc_init: int = 0x555555,
                 first_sample_index: int = 0, cap: int = 0):
    a = _f32(iq)
    n = a.size // 2
    cap = cap or max(64, n // 512)
    out = np.zeros(cap, dtype=PKT_DTYPE)
    n_out = C.c_uint64(0)
    n_hits = C.c_uint64(0)
    hits = np.zeros(cap * 4, dtype=np.uint64)
    rc = lib().oracle_btle_segment(_p(a, C.c_float), n, first_sample_index, channel, aa, crc_init,
                                   out.ctypes.data_as(C.c_void_p), cap, C.byref(n_out),
                                   _p(hits, C.c_uint64), hits.size, C.byref(n_hits))
    assert rc == 0, rc
    return out[:n_out.value], hits[:min(n_hits.value, hits.size)]


def from_int(a: np.ndarray) -> np.ndarray:
    """Interleaved int8 / int16 IQ -> the cf32 capture it stands for (include/snout_rx.h
    SNOUT_FMT_SC8 / SC16: v * 2^-7 / v * 2^-15, exact in float32).  The oracle's definition of the
    integer input formats is: run the cf32 algorithm on this."""
    a = np.ascontiguousarray(a)
    scale = {np.dtype(np.int8): 2.0 ** -7, np.dtype(np.int16): 2.0 ** -15}[a.dtype]
    return (a.astype(np.float32) * np.float32(scale)).astype(np.float32)


# ---- Zigbee ------------------------------------------------------------------------------------
def zb_chip_map() -> np.ndarray:
    return np.array(lib().oracle_zb_chip_map()[:16], dtype=np.uint32)


def zb_mmse_taps() -> np.ndarray:
    return np.array(lib().oracle_zb_mmse_taps()[:129 * 8], dtype=np.float32).reshape(129, 8)


def fast_atan2f(y: float, x: float) -> float:
    return float(lib().oracle_fast_atan2f(y, x))


def crc16_154(data: bytes) -> int:
    a = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    return int(lib().oracle_crc16_154(_p(a, C.c_uint8), a.size))


def zb_discrim(iq: np.ndarray) -> np.ndarray:
    a = _f32(iq)
    n = a.size // 2
    d = np.zeros(max(n, 1), dtype=np.float32)
    lib().oracle_zb_discrim(_p(a, C.c_float), n, _p(d, C.c_float))
    return d[:n]


def zb_auto_shape(n_channels: int = 16):
    """The product's default lane shape (cfg.zb_core = cfg.zb_warmup = 0; snout_zigbee_lane_shape(n_channels)): one shape per
    kind of handle whatever the size of a call (ABI 3); ABI 4: warm-up 3072 for wideband handles, 1024 for narrowband ones."""
    return (6144, 1024) if n_channels <= 1 else (6144, 3072)


def zigbee_segment(iq: np.ndarray, channel: int = 11, threshold: int = 10, core: int = 0,
                   warmup: int = 0, first_sample_index: int = 0, cap: int = 0) -> np.ndarray:
    """core = warmup = 0: the product's default for a narrowband handle; core alone: warm-up 512."""
    a = _f32(iq)
    n = a.size // 2
    if core == 0 and warmup == 0:
        core, warmup = zb_auto_shape(1)
    core, warmup = core or 2048, warmup or 512
    cap = cap or max(64, n // 512)
    out = np.zeros(cap, dtype=PKT_DTYPE)
    n_out = C.c_uint64(0)
    rc = lib().oracle_zigbee_segment(_p(a, C.c_float), n, first_sample_index, channel, threshold,
                                     core, warmup, out.ctypes.data_as(C.c_void_p), cap,
                                     C.byref(n_out))
    assert rc == 0, rc
    return out[:n_out.value]


def zigbee_lane_soft(iq: np.ndarray, lane: int = 0, core: int = 2048, warmup: int = 512,
                     threshold: int = 10, cap: int = 1 << 17):
    a = _f32(iq)
    n = a.size // 2
    z = np.zeros(cap, dtype=np.float32)
    chips = np.zeros(cap, dtype=np.float32)
    nc = C.c_uint64(0)
    rc = lib().oracle_zigbee_lane_soft(_p(a, C.c_float), n, core, warmup, lane, threshold,
                                       _p(z, C.c_float), _p(chips, C.c_float), cap, C.byref(nc))
    assert rc == 0
    return z, chips[:min(nc.value, cap)]


# ---- polyphase channelizer / wideband ------------------------------------------------------------
def pfb_nout(n: int, M: int) -> int:
    return int(lib().oracle_pfb_nout(n, M))


def pfb_proto(M: int) -> np.ndarray:
    return np.array(lib().oracle_pfb_proto(M)[:M * 16], dtype=np.float32)


def pfb(iq: np.ndarray, M: int, block_order: bool = False, legacy_fft: bool = False) -> np.ndarray:
    """-> complex64 [M, n_out].  block_order: the term order of the experimental matrix-pipe FIR (SNOUT_PFB_IMPL=mfma).
    legacy_fft (M = 40): the Cooley-Tukey FFT with twiddles of rounds 1-3 instead of the shipped prime-factor form -- the
    specification of the A/B partners pfb.hip / pfb_mfma.hip (libsnout_rx_ab.so) only."""
    a = _f32(iq)
    n = a.size // 2
    no = pfb_nout(n, M)
    y = np.zeros((M, max(no, 1) * 2), dtype=np.float32)
    fn = lib().oracle_pfb_block_order if block_order else lib().oracle_pfb
    lib().oracle_pfb_legacy_fft(1 if legacy_fft else 0)
    try:
        rc = fn(_p(a, C.c_float), n, M, _p(y, C.c_float), max(no, 1))
    finally:
        lib().oracle_pfb_legacy_fft(0)
    assert rc == 0
    return y.view(np.complex64)[:, :no]


def btle_bin_channel(b: int) -> int:
    return int(lib().oracle_btle_bin_channel(b))


def zigbee_bin_channel(b: int) -> int:
    return int(lib().oracle_zigbee_bin_channel(b))


def wideband_segment(iq: np.ndarray, proto: int, first_sample_index: int = 0, aa: int = 0x8E89BED6,
                     crc_init: int = 0x555555, threshold: int = 10, core: int = 0,
                     warmup: int = 0, cap: int = 0) -> np.ndarray:
    a = _f32(iq)
    n = a.size // 2
    if proto == 1 and core == 0 and warmup == 0:        # the product's default
        core, warmup = zb_auto_shape(16)
    core, warmup = core or 2048, warmup or 512
    cap = cap or max(256, n // 128)
    out = np.zeros(cap, dtype=PKT_DTYPE)
    n_out = C.c_uint64(0)
    rc = lib().oracle_wideband_segment(_p(a, C.c_float), n, first_sample_index, proto, aa, crc_init,
                                       threshold, core, warmup, out.ctypes.data_as(C.c_void_p), cap,
                                       C.byref(n_out))
    assert rc == 0, rc
    return out[:n_out.value]


# ---- threads (bench.py cpu_baseline legs) ------------------------------------------------------------
def set_threads(n: int) -> int:
    """OpenMP threads used by oracle_pfb and the parallel receivers (1 = the scalar port)."""
    return int(lib().oracle_set_threads(int(n)))


def hw_threads() -> int:
    return int(lib().oracle_hw_threads())


def narrowband_parallel(iq: np.ndarray, proto: int, channel: int, seg: int = 1 << 22, overlap: int = 0,
                        aa: int = 0x8E89BED6, crc_init: int = 0x555555, threshold: int = 10,
                        core: int = 2048, warmup: int = 512, cap: int = 0) -> np.ndarray:
    """Overlapping segments of one single-channel capture, one OpenMP task each (all-cores leg)."""
    a = _f32(iq)
    n = a.size // 2
    overlap = overlap or (1600 if proto == 0 else 17024 + 2048)
    cap = cap or max(256, n // 256)
    out = np.zeros(cap, dtype=PKT_DTYPE)
    n_out = C.c_uint64(0)
    rc = lib().oracle_narrowband_parallel(_p(a, C.c_float), n, proto, channel, aa, crc_init, threshold,
                                          core, warmup, seg, overlap, out.ctypes.data_as(C.c_void_p), cap,
                                          C.byref(n_out))
    assert rc == 0, rc
    return out[:n_out.value]


def wideband_parallel(iq: np.ndarray, proto: int, seg: int, aa: int = 0x8E89BED6, crc_init: int = 0x555555,
                      threshold: int = 10, core: int = 2048, warmup: int = 512, cap: int = 0) -> np.ndarray:
    """Overlapping segments of one wideband capture, one OpenMP task each (all-cores leg)."""
    a = _f32(iq)
    n = a.size // 2
    cap = cap or max(256, n // 128)
    out = np.zeros(cap, dtype=PKT_DTYPE)
    n_out = C.c_uint64(0)
    rc = lib().oracle_wideband_parallel(_p(a, C.c_float), n, proto, aa, crc_init, threshold, core, warmup,
                                        seg, out.ctypes.data_as(C.c_void_p), cap, C.byref(n_out))
    assert rc == 0, rc
    return out[:n_out.value]
