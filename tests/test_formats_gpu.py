"""GPU parity: integer input sample formats (include/snout_rx.h SNOUT_FMT_SC8 / SC16).

An sc8 / sc16 capture stands for the cf32 capture v * 2^-7 / v * 2^-15 (exact in float32), so the
check is: records from the integer capture through the C ABI == the CPU oracle on the converted
capture, bit for bit, on all four receive paths, from host and from device memory, for ragged
lengths and the integer extremes (-128 / -32768)."""
import numpy as np
import pytest

from snout_amd import synth
from snout_amd._ffi import STAGE_BTLE_BITS, STAGE_CHAN_IQ, STAGE_ZB_DISCRIM

pytestmark = pytest.mark.gpu

FIELDS = ("sample_index", "proto", "channel", "len", "crc_ok", "lqi", "pdu_type", "flags", "aux")


def _same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(a["bytes"], b["bytes"])


def _rx(**kw):
    from snout_amd.rx import SnoutRx
    return SnoutRx(**kw)


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("n,seed", [(1 << 18, 31), ((1 << 17) + 1234, 32), (16384 * 3 + 5, 33)])
def test_btle_narrowband(oracle, fmt, n, seed):
    import torch
    x, truth = synth.btle_capture(n, channel=37, seed=seed, mean_gap=6000.0)
    q = synth.quantize(x, fmt)
    want, hits = oracle.btle_segment(oracle.from_int(q), channel=37, first_sample_index=99)
    with _rx(proto=0, channel=37, sample_format=fmt) as rx:
        got = rx.process(q, first_sample_index=99)                       # host input
        _same(got, want)
        bits = rx.soft(STAGE_BTLE_BITS, 0).astype(np.uint8)
        assert np.array_equal(bits, oracle.btle_bits(oracle.from_int(q)))
        got_d = rx.process(torch.from_numpy(q).cuda(), first_sample_index=99)   # device input
        _same(got_d, want)
    ok = {bytes(p["bytes"][:p["len"] - 3]) for p in got if p["crc_ok"]}
    assert sum(t.payload in ok for t in truth) >= len(truth) - 1 and len(truth) > 3


@pytest.mark.parametrize("fmt", [1, 2])
def test_btle_integer_extremes_and_noise(oracle, fmt):
    """Uniform random integers over the whole range, including -128 / -32768 (whose negation does
    not exist) and runs of zeros: every hard bit equals the oracle's."""
    rng = np.random.default_rng(5 + fmt)
    info = np.iinfo(np.int8 if fmt == 1 else np.int16)
    q = rng.integers(info.min, info.max + 1, size=2 * 70001, dtype=info.dtype)
    q[1000:3000] = 0
    q[5000:5064] = info.min
    q[7000:7064:2] = info.min
    q[7001:7064:2] = info.max
    with _rx(proto=0, channel=12, sample_format=fmt) as rx:
        got = rx.process(q)
        bits = rx.soft(STAGE_BTLE_BITS, 0).astype(np.uint8)
    f = oracle.from_int(q)
    assert np.array_equal(bits, oracle.btle_bits(f))
    _same(got, oracle.btle_segment(f, channel=12)[0])


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("n,seed", [(1 << 18, 41), ((1 << 17) + 777, 42)])
def test_zigbee_narrowband(oracle, fmt, n, seed):
    import torch
    x, truth = synth.zigbee_capture(n, channel=15, seed=seed, mean_gap=9000.0)
    q = synth.quantize(x, fmt)
    f = oracle.from_int(q)
    want = oracle.zigbee_segment(f, channel=15, first_sample_index=7)
    with _rx(proto=1, channel=15, sample_format=fmt) as rx:
        got = rx.process(q, first_sample_index=7)
        _same(got, want)
        d = rx.soft(STAGE_ZB_DISCRIM, 0)
        assert np.array_equal(d, oracle.zb_discrim(f))      # same operations in the same order
        _same(rx.process(torch.from_numpy(q).cuda(), first_sample_index=7), want)
    ok = {bytes(p["bytes"][:p["len"]]) for p in got if p["crc_ok"]}
    assert sum(t.payload in ok for t in truth) >= 0.9 * len(truth) and len(truth) > 3


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("proto,M,n", [(0, 40, 40 * 30000 + 13), (1, 16, 16 * 70000 + 5)])
def test_wideband(oracle, fmt, proto, M, n):
    bins = [3, 17, 28] if proto == 0 else [2, 9]
    x, truth = synth.wideband_capture(proto, n, seed=50 + fmt, bins=bins, mean_gap=5000.0)
    q = synth.quantize(x, fmt)
    f = oracle.from_int(q)
    want = oracle.wideband_segment(f, proto=proto)
    with _rx(proto=proto, n_channels=M, sample_format=fmt) as rx:        # fused kernels
        _same(rx.process(q), want)
    with _rx(proto=proto, n_channels=M, sample_format=fmt, keep_channel_iq=True) as rx:
        _same(rx.process(q), want)
        y = oracle.pfb(f, M)
        for slot in (bins[0], M - 1):
            got_y = rx.soft(STAGE_CHAN_IQ, slot).view(np.complex64)
            assert np.array_equal(got_y, y[slot])
    ok = sum(1 for p in want if p["crc_ok"])
    assert ok >= 0.8 * len(truth) and len(truth) > 5


def test_wrong_dtype_is_refused():
    with _rx(proto=0, channel=37, sample_format=1) as rx:
        with pytest.raises(TypeError):
            rx.process(np.zeros(1024, dtype=np.float32))
    from snout_amd._ffi import SnoutError
    with pytest.raises(SnoutError):
        _rx(proto=0, channel=37, sample_format=3)


def test_hackrf_style_file_through_the_btle_rx_child(tmp_path):
    """The cfg #1 recording re-quantised to int8 IQ (what `hackrf_transfer -r` writes and upstream
    btle_rx consumes): `snout_amd.cli btle-rx --format sc8` prints the same 8 lines' fields."""
    import json
    import os
    import subprocess
    import sys
    from snout_amd.message import BtleMessage
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    x = np.fromfile(os.path.join(gold, "btle_ch37_4msps.cf32"), dtype=np.complex64)
    path = tmp_path / "btle_ch37.sc8"
    synth.quantize(x, 1).tofile(path)
    out = subprocess.run([sys.executable, "-m", "snout_amd.cli", "btle-rx", "-c", "37", "-g", "6",
                          "-a", "8e89bed6", "-k", "555555", "--iq", str(path), "--format", "sc8"],
                         cwd=root, capture_output=True, timeout=300, check=True)
    msgs = [m for m in (BtleMessage.fromraw(ln) for ln in out.stdout.splitlines(keepends=True)) if m]
    truth = json.load(open(os.path.join(gold, "btle_ch37_truth.json")))
    assert len(msgs) == 8
    for m, t in zip(msgs, truth):
        pdu = bytes.fromhex(t["pdu"])
        assert m.sender == pdu[2:8][::-1].hex() and m.payload_hex == pdu[8:].hex()


def test_sc16_file_through_zigbee_scan(tmp_path):
    import json
    import os
    from snout_amd.scan import FileSource, ZigbeeScan
    gold = os.path.join(os.path.dirname(__file__), "golden")
    x = np.fromfile(os.path.join(gold, "zigbee_ch15_4msps.cf32"), dtype=np.complex64)
    path = tmp_path / "zb.sc16"
    synth.quantize(x, 2).tofile(path)
    scan = ZigbeeScan(channels=[15], source=FileSource(str(path), sample_format=2), timeout=None)
    msgs = scan.run()
    want = json.load(open(os.path.join(gold, "zigbee_ch15_expected.json")))
    sent = [f["psdu"] for f in want["sent"]]
    got = {m.mpdu.hex() for m in msgs}
    assert len(sent) > 3 and sum(s in got for s in sent) >= len(sent) - 1


def test_sharded_wideband_scan_on_int8_input():
    """Segments of an sc8 wideband capture through ShardedScan == the capture processed whole."""
    import torch
    from snout_amd.sharded import ShardedScan
    x, truth = synth.wideband_capture(0, 40 * 120000, seed=61, bins=[2, 11, 31], mean_gap=5000.0)
    q = torch.from_numpy(synth.quantize(x, 1)).cuda()
    with _rx(proto=0, n_channels=40, sample_format=1) as rx:
        whole = rx.process(q)
    sc = ShardedScan(proto=0, n_channels=40, seg_len=40 * 30000, sample_format=1)
    got = sc.run(len(x), lambda a, b: q[2 * a:2 * b])
    sc.close()
    key = lambda r: {(int(p["channel"]), int(p["sample_index"]), bytes(p["bytes"][:p["len"]])) for p in r if p["crc_ok"]}
    assert key(got) == key(whole) and len(key(whole)) >= 0.9 * len(truth)
