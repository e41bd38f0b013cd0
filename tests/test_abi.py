"""The drop-in boundary: libsnout_rx.so loads, exports every symbol include/snout_rx.h declares,
struct layouts match, and (without a GPU) fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    """Every function any header under include/ declares (snout_rx.h: the receive-path ABI; snout_bench.h: the
    measurement aid bench.py uses, kept out of the drop-in boundary)."""
    names = set()
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(snout_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_bench_aid_is_not_in_the_product_header():
    src = open(os.path.join(ROOT, "include", "snout_rx.h")).read()
    assert "hbm_read" not in src and "snout_bench_hbm_read_gbps" in open(os.path.join(ROOT, "include", "snout_bench.h")).read()


def test_library_exports_every_declared_symbol():
    from snout_amd import _ffi
    lib = _ffi.load()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared under include/ but not exported"
    assert set(names) == set(_ffi.EXPORTS)


def test_struct_layouts():
    from snout_amd import _ffi
    assert _ffi.PKT_DTYPE.itemsize == 160
    assert _ffi.PKT_DTYPE.fields["bytes"][1] == 24
    assert C.sizeof(_ffi.RxCfg) == 16 * 4
    assert C.sizeof(_ffi.RxProf) == 24 + 48


def test_strerror_and_channel_plans():
    from snout_amd import _ffi
    lib = _ffi.load()
    assert lib.snout_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.snout_strerror(-2)
    # snout/modulations/Zigbee/hackrf/Zigbee_rx/top_block.py:56 : 1e6*(2400+5*(ch-10))
    for ch in range(11, 27):
        assert lib.snout_zigbee_center_hz(ch) == 1e6 * (2400 + 5 * (ch - 10))
    assert lib.snout_btle_center_hz(37) == 2402e6
    assert lib.snout_btle_center_hz(38) == 2426e6
    assert lib.snout_btle_center_hz(39) == 2480e6
    assert lib.snout_btle_center_hz(0) == 2404e6 and lib.snout_btle_center_hz(11) == 2428e6
    rf = [lib.snout_btle_rf_to_channel(k) for k in range(40)]
    assert sorted(rf) == list(range(40)) and rf[0] == 37 and rf[12] == 38 and rf[39] == 39
    for k, ch in enumerate(rf):
        assert lib.snout_btle_center_hz(ch) == (2402 + 2 * k) * 1e6
    assert lib.snout_btle_rf_to_channel(40) == -1


def test_no_silent_cpu_fallback():
    """Without a GPU the product must refuse to run (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from snout_amd._ffi import SnoutError
    from snout_amd.rx import SnoutRx
    with pytest.raises(SnoutError) as e:
        SnoutRx(proto=0, channel=37)
    assert e.value.code == -2


def test_bad_arguments_are_rejected():
    from snout_amd import _ffi
    lib = _ffi.load()
    h = C.c_void_p()
    assert lib.snout_rx_create(None, C.byref(h)) == -1
    cfg = _ffi.RxCfg(abi_version=99)
    assert lib.snout_rx_create(C.byref(cfg), C.byref(h)) == -1
    n = C.c_uint64()
    assert lib.snout_rx_process(None, None, 0, 0, None, 0, C.byref(n)) == -1
    rec = np.zeros(1, dtype=_ffi.PKT_DTYPE)
    buf = C.create_string_buffer(16)
    assert lib.snout_btle_format_line(rec.ctypes.data_as(C.c_void_p), 4e6, 0.0, 0, 0, buf, 16) == -1


def test_cfg_layout_matches_the_header():
    """snout_rx_cfg as ctypes sees it == the C struct: compiled probe of offsetof / sizeof."""
    import ctypes as C
    import os
    import subprocess
    import tempfile
    from snout_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "snout_rx.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(snout_rx_cfg), offsetof(snout_rx_cfg, device),
           offsetof(snout_rx_cfg, flags), offsetof(snout_rx_cfg, sample_format),
           offsetof(snout_rx_cfg, batch_segments), offsetof(snout_rx_cfg, reserved_cus), sizeof(snout_pkt));
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), os.path.join(d, "p.c"), "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [C.sizeof(_ffi.RxCfg), _ffi.RxCfg.device.offset, _ffi.RxCfg.flags.offset,
            _ffi.RxCfg.sample_format.offset, _ffi.RxCfg.batch_segments.offset, _ffi.RxCfg.reserved_cus.offset,
            _ffi.PKT_DTYPE.itemsize]
    assert got == want == [64, 44, 48, 52, 56, 60, 160]


def test_lane_shape_rule_is_the_oracles():
    """The default 802.15.4 lane shape is ONE shape whatever the size of the call (ABI 3: the records of a capture must not
    depend on how it is cut into submissions); the checker (oracle_py.zb_auto_shape) must run the shape the product runs."""
    from snout_amd import _ffi
    from oracle import oracle_py
    lib = _ffi.load()
    c, w = C.c_uint32(0), C.c_uint32(0)
    for total in (16, 2, 1 << 20, 10 ** 9, 1 << 40):
        lib.snout_zigbee_lane_shape(C.c_uint64(total), C.byref(c), C.byref(w))          # (any n_channels > 1 is a wideband handle)
        assert (c.value, w.value) == oracle_py.zb_auto_shape(max(2, total)) == (6144, 3072), total
    for nch in (0, 1):                                  # a narrowband handle keeps ABI 3's warm-up
        lib.snout_zigbee_lane_shape(C.c_uint64(nch), C.byref(c), C.byref(w))
        assert (c.value, w.value) == oracle_py.zb_auto_shape(nch) == (6144, 1024)


def test_abi_version_and_record_flags_of_the_header():
    """ABI 3 (round 5): one default 802.15.4 lane shape and the frame repair's record flag; ABI 4 (round 6): that shape's warm-up.  The header, the binding and the
    library agree on the version; the two 802.15.4 record flags are distinct bits that do not collide with the BTLE
    TxAdd / RxAdd bits."""
    import os
    import re
    from snout_amd import _ffi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "snout_rx.h")).read()
    ver = int(re.search(r"#define SNOUT_ABI_VERSION (\d+)u", hdr).group(1))
    assert ver == _ffi.ABI_VERSION == _ffi.load().snout_abi_version() == 4
    seam = int(re.search(r"#define SNOUT_PKT_ZB_SEAM_DISAGREED (0x[0-9a-fA-F]+)u", hdr).group(1), 16)
    rep = int(re.search(r"#define SNOUT_PKT_ZB_REPAIRED (0x[0-9a-fA-F]+)u", hdr).group(1), 16)
    assert (seam, rep) == (4, 8) and not (seam | rep) & 3
