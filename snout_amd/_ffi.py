"""ctypes binding of libsnout_rx.so (include/snout_rx.h).

The library is the product: there is no Python or CPU fallback. A missing library raises
ImportError at load time; a missing gfx950 device makes ``snout_rx_create`` fail with
``SNOUT_ENODEV`` which is raised as :class:`SnoutError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNOUT_RX_LIB") or os.path.join(_HERE, "lib", "libsnout_rx.so")   # override: A/B builds

ABI_VERSION = 4
PROTO_BTLE, PROTO_ZIGBEE = 0, 1
CFG_KEEP_CHANNEL_IQ, CFG_RECORDS_ON_DEVICE = 1, 2
STAGE_BTLE_BITS, STAGE_CHAN_IQ, STAGE_ZB_DISCRIM, STAGE_ZB_DCREMOVED, STAGE_ZB_CHIPS = range(5)

EXPORTS = [
    "snout_rx_create", "snout_rx_destroy", "snout_rx_process", "snout_rx_process_dev",
    "snout_rx_submit_dev", "snout_rx_collect", "snout_rx_collect_view", "snout_rx_last_records_dev",
    "snout_host_alloc", "snout_host_free", "snout_rx_soft", "snout_rx_profile",
    "snout_rx_profile_history", "snout_btle_format_line", "snout_rftap_encap",
    "snout_zigbee_center_hz", "snout_btle_center_hz", "snout_btle_rf_to_channel",
    "snout_strerror", "snout_last_error", "snout_abi_version", "snout_bench_hbm_read_gbps",
    "snout_rx_pack_last_records", "snout_rx_submit_batch_dev", "snout_rx_poll", "snout_zigbee_lane_shape",
    "snout_records_dedup", "snout_records_dedup_workspace",
]


class SnoutError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"libsnout_rx error {code}: {detail}")


class RxCfg(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("proto", C.c_uint32), ("n_channels", C.c_uint32),
                ("taps_per_branch", C.c_uint32), ("channel", C.c_uint32),
                ("access_addr", C.c_uint32), ("crc_init", C.c_uint32),
                ("chip_threshold", C.c_uint32), ("zb_core", C.c_uint32), ("zb_warmup", C.c_uint32),
                ("max_hits", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("sample_format", C.c_uint32), ("batch_segments", C.c_uint32), ("reserved_cus", C.c_uint32)]


class RxProf(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_dominant", C.c_float),
                ("dominant_launches", C.c_uint32), ("n_hits", C.c_uint32),
                ("bytes_algorithmic", C.c_uint64), ("dominant_name", C.c_char * 48)]


PKT_DTYPE = np.dtype([("sample_index", "<u8"), ("proto", "<u4"), ("channel", "<u2"),
                      ("len", "<u2"), ("crc_ok", "u1"), ("lqi", "u1"), ("pdu_type", "u1"),
                      ("flags", "u1"), ("aux", "<u4"), ("bytes", "u1", (136,))])
assert PKT_DTYPE.itemsize == 160

_lib = None


def load() -> C.CDLL:
    """Load libsnout_rx.so. If torch is installed it is imported first so that both share one
    HIP runtime (same SONAME libamdhip64.so.7) and device pointers / streams can be exchanged."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C snout_amd/csrc`. snout_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads the bundled libamdhip64 first)
    except Exception:  # pragma: no cover - torch is optional for pure-ctypes use
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64 = C.c_void_p, C.c_uint64
    lib.snout_rx_create.argtypes = [C.POINTER(RxCfg), C.POINTER(vp)]
    lib.snout_rx_create.restype = C.c_int
    lib.snout_rx_destroy.argtypes = [vp]
    lib.snout_rx_destroy.restype = None
    lib.snout_rx_process.argtypes = [vp, vp, u64, u64, vp, u64, C.POINTER(u64)]
    lib.snout_rx_process.restype = C.c_int
    lib.snout_rx_process_dev.argtypes = [vp, vp, u64, u64, vp, vp, u64, C.POINTER(u64)]
    lib.snout_rx_process_dev.restype = C.c_int
    lib.snout_rx_submit_dev.argtypes = [vp, vp, u64, u64, vp]
    lib.snout_rx_submit_dev.restype = C.c_int
    lib.snout_rx_submit_batch_dev.argtypes = [vp, C.POINTER(vp), C.c_uint32, u64, C.POINTER(u64), C.POINTER(u64), vp]
    lib.snout_rx_submit_batch_dev.restype = C.c_int
    lib.snout_rx_poll.argtypes = [vp]
    lib.snout_rx_poll.restype = C.c_int
    lib.snout_rx_collect.argtypes = [vp, vp, u64, C.POINTER(u64)]
    lib.snout_rx_collect.restype = C.c_int
    lib.snout_rx_collect_view.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.snout_rx_collect_view.restype = C.c_int
    lib.snout_rx_last_records_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.snout_rx_last_records_dev.restype = C.c_int
    lib.snout_host_alloc.argtypes = [C.c_size_t]
    lib.snout_host_alloc.restype = vp
    lib.snout_host_free.argtypes = [vp]
    lib.snout_host_free.restype = None
    lib.snout_rx_soft.argtypes = [vp, C.c_uint32, C.c_uint32, vp, u64, C.POINTER(u64)]
    lib.snout_rx_soft.restype = C.c_int
    lib.snout_rx_profile.argtypes = [vp, C.POINTER(RxProf)]
    lib.snout_rx_profile.restype = C.c_int
    lib.snout_rx_profile_history.argtypes = [vp, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.snout_rx_profile_history.restype = C.c_int
    lib.snout_btle_format_line.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                           C.c_char_p, C.c_size_t]
    lib.snout_btle_format_line.restype = C.c_int
    lib.snout_rftap_encap.argtypes = [vp, vp, C.c_size_t]
    lib.snout_rftap_encap.restype = C.c_int
    lib.snout_zigbee_center_hz.argtypes = [C.c_uint32]
    lib.snout_zigbee_center_hz.restype = C.c_double
    lib.snout_btle_center_hz.argtypes = [C.c_uint32]
    lib.snout_btle_center_hz.restype = C.c_double
    lib.snout_btle_rf_to_channel.argtypes = [C.c_uint32]
    lib.snout_btle_rf_to_channel.restype = C.c_int32
    lib.snout_strerror.argtypes = [C.c_int]
    lib.snout_strerror.restype = C.c_char_p
    lib.snout_last_error.argtypes = []
    lib.snout_last_error.restype = C.c_char_p
    lib.snout_rx_pack_last_records.argtypes = [vp, vp, u64, C.c_uint32, u64, u64, vp, vp, C.POINTER(u64)]
    lib.snout_rx_pack_last_records.restype = C.c_int
    lib.snout_records_dedup_workspace.argtypes = [C.c_uint32, u64]
    lib.snout_records_dedup_workspace.restype = C.c_size_t
    lib.snout_records_dedup.argtypes = [vp, C.c_uint32, C.c_uint32, u64, vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_size_t, vp]
    lib.snout_records_dedup.restype = C.c_int
    lib.snout_bench_hbm_read_gbps.argtypes = [vp, u64, C.c_uint32, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.snout_bench_hbm_read_gbps.restype = C.c_int
    lib.snout_abi_version.argtypes = []
    lib.snout_abi_version.restype = C.c_uint32
    if lib.snout_abi_version() != ABI_VERSION:
        raise ImportError("libsnout_rx.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, allow_overflow: bool = False) -> int:
    if rc == 0 or (allow_overflow and rc == -5):
        return rc
    lib = load()
    detail = (lib.snout_last_error() or b"").decode(errors="replace")
    raise SnoutError(rc, f"{lib.snout_strerror(rc).decode()}: {detail}")
