"""GPU parity: libsnout_rx.so BTLE path (through the C ABI) vs the CPU oracle, bit-exact.

Counterpart of the checks the reference has no tests for: demodulation, access-address
correlation, de-whitening, CRC of the path behind snout/util/btle.py:53-76.
"""
import numpy as np
import pytest

from snout_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rx():
    from snout_amd.rx import SnoutRx
    r = SnoutRx(proto=0, channel=37)
    yield r
    r.close()


def _same_packets(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("sample_index", "proto", "channel", "len", "crc_ok", "pdu_type", "flags", "aux"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(a["bytes"], b["bytes"])


@pytest.mark.parametrize("n,seed", [(1 << 16, 1), (1 << 20, 2), ((1 << 20) + 12345, 3),
                                     (3 * 16384 + 7, 4), (16384, 5), (16383, 6), (16385, 7)])
def test_packets_match_oracle(rx, oracle, n, seed):
    x, truth = synth.btle_capture(n, seed=seed, mean_gap=6000.0)
    got = rx.process(x, first_sample_index=1000)
    want, _ = oracle.btle_segment(x, first_sample_index=1000)
    _same_packets(got, want)
    assert len(got) >= len(truth) > 0
    # round trip: every generated PDU comes back with a good CRC
    ok = {bytes(p["bytes"][:p["len"] - 3]) for p in got if p["crc_ok"]}
    assert all(t.payload in ok for t in truth)


def test_hard_bits_match_oracle(rx, oracle):
    x, _ = synth.btle_capture(200_000, seed=11, mean_gap=5000.0)
    rx.process(x)
    from snout_amd._ffi import STAGE_BTLE_BITS
    bits = rx.soft(STAGE_BTLE_BITS)
    want = oracle.btle_bits(x)
    assert bits.size == want.size
    assert np.array_equal(bits.astype(np.uint8), want)


@pytest.mark.parametrize("n", [0, 1, 4, 5, 123, 124, 125, 255, 256, 257, 1503])
def test_tiny_segments(rx, oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    got = rx.process(x)
    want, _ = oracle.btle_segment(x) if n else (np.zeros(0, dtype=got.dtype), None)
    _same_packets(got, want)


def test_noise_only_and_nonfinite(rx, oracle):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(1 << 18) + 1j * rng.standard_normal(1 << 18)).astype(np.complex64)
    x[1000] = np.nan
    x[2000] = np.inf
    x[3000:3100] = 0
    got = rx.process(x)
    want, _ = oracle.btle_segment(x)
    _same_packets(got, want)
    from snout_amd._ffi import STAGE_BTLE_BITS
    assert np.array_equal(rx.soft(STAGE_BTLE_BITS).astype(np.uint8), oracle.btle_bits(x))


def test_dense_false_hits_follow_sequential_rule(rx, oracle):
    """Back-to-back packets and packets truncated by the segment end: the parallel cluster
    resolution must reproduce the sequential resume-after-packet rule."""
    x, truth = synth.btle_capture(1 << 18, seed=21, mean_gap=40.0, sigma=0.02)
    for cut in (len(x), len(x) - 777, truth[-1].sample_index + 300, truth[-1].sample_index + 129):
        got = rx.process(x[:cut])
        want, _ = oracle.btle_segment(x[:cut])
        _same_packets(got, want)


def test_device_resident_input(rx, oracle):
    import torch
    x, _ = synth.btle_capture(1 << 20, seed=31)
    t = torch.from_numpy(x.view(np.float32)).cuda()
    got = rx.process(t)
    want, _ = oracle.btle_segment(x)
    _same_packets(got, want)
    p = rx.profile()
    assert p.dominant_name == "btle_demod_corr" and p.ms_dominant > 0


def test_cfg1_fixture_through_scan(tmp_path, oracle):
    """cfg #1 plumbing: recorded ch37 file -> BtleScan -> btle_rx lines -> parser; 8/8 CRC0."""
    import json
    import os
    from snout_amd.scan import BtleScan, FileSource
    gold = os.path.join(os.path.dirname(__file__), "golden")
    dump = tmp_path / "btle.b"
    scan = BtleScan(channels=[37], source=FileSource(os.path.join(gold, "btle_ch37_4msps.cf32")),
                    timeout=None, filename=str(dump), t0_epoch=1567108496.0)
    seen = []
    scan.events.on("btle.packet-received", lambda message: seen.append(message.sender))
    msgs = scan.run()
    truth = json.load(open(os.path.join(gold, "btle_ch37_truth.json")))
    assert len(msgs) == 8 == len(seen)
    for m, t in zip(msgs, truth):
        pdu = bytes.fromhex(t["pdu"])
        assert m.sender == pdu[2:8][::-1].hex() and m.payload_hex == pdu[8:].hex()
        assert m.channel == "37"
    cases = json.load(open(os.path.join(gold, "btle_lines.json")))
    assert dump.read_bytes().decode() == "".join(c["line"] for c in cases[5:13])
    # packet threshold stops the scan early, like check_stop (btle.py:111-122)
    scan2 = BtleScan(channels=[37], source=FileSource(os.path.join(gold, "btle_ch37_4msps.cf32")),
                     timeout=None, packet_threshold=3)
    assert len(scan2.run()) == 3


def test_pipelined_submit_collect_matches_sync(oracle):
    """Two segments in flight: records equal the synchronous path, in submission order."""
    import torch
    from snout_amd.rx import SnoutRx
    from snout_amd._ffi import SnoutError
    segs = []
    for seed in (41, 42, 43, 44):
        x, _ = synth.btle_capture(1 << 19, seed=seed, mean_gap=7000.0)
        segs.append((x, torch.from_numpy(x.view(np.float32)).cuda()))
    with SnoutRx(proto=0, channel=37) as rx:
        got = []
        rx.submit(segs[0][1], first_sample_index=0)
        for i in range(1, len(segs)):
            rx.submit(segs[i][1], first_sample_index=i << 19)
            got.append(rx.collect())
        with pytest.raises(SnoutError):
            for _ in range(4):
                rx.submit(segs[0][1])
        rest = [rx.collect(), rx.collect(), rx.collect()]
        got.append(rest[0]) if len(got) < len(segs) else None
        with pytest.raises(SnoutError):
            rx.collect()
        for i, (x, _) in enumerate(segs):
            want, _h = oracle.btle_segment(x, first_sample_index=i << 19)
            _same_packets(got[i], want)
        # a view stays valid across one more submit
        rx.submit(segs[1][1]); v = rx.collect(copy=False); keep = v.copy()
        rx.submit(segs[2][1]); assert np.array_equal(v["bytes"], keep["bytes"]); rx.collect()


def test_async_record_gather_over_rccl_world1():
    """The nccl(=RCCL) code path of the pipelined gather on one GPU (world_size 1)."""
    import socket
    import torch
    import torch.distributed as dist
    from snout_amd import dist as sdist
    from snout_amd.rx import SnoutRx
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        x, _ = synth.btle_capture(1 << 20, seed=51, mean_gap=5000.0)
        t = torch.from_numpy(x.view(np.float32)).cuda()
        g = sdist.AsyncRecordGather(torch.device("cuda", 0), width=80)
        with SnoutRx(proto=0, channel=37) as rx:
            want = rx.process(t)
            outs = []
            for i in range(6):
                g.sync_uploads()
                rx.submit(t)
                if i:
                    pk = rx.collect(copy=False)
                    if len(g.inflight) == 2:
                        outs.append(g.finish())
                    # odd steps: upload from the pinned host view; even steps: packed straight from
                    # the device copy of the records (snout_rx_last_records_dev)
                    dev, n_dev = rx.last_records_device()
                    assert n_dev == len(pk)
                    g.start(pk, dev if i % 2 == 0 else 0)
            g.start(rx.collect(copy=False)) if len(g.inflight) < 2 else None
            while g.inflight:
                outs.append(g.finish())
        assert len(outs) >= 5
        for o in outs:
            assert o.dtype.itemsize == 80 and len(o) == len(want)
            assert np.array_equal(o["sample_index"], want["sample_index"])
            assert np.array_equal(o["bytes"], want["bytes"][:, :56])
    finally:
        dist.destroy_process_group()


def test_btle_rx_child_process_drop_in():
    """The reference drives `btle_rx` as a child and parses its stdout (snout/util/btle.py:53-76,
    snout/core/pcontroller.py).  `snout_amd.cli btle-rx` takes the same argv and prints the same
    lines: run it as a real subprocess on the cfg #1 recording, like tests/test_util_pcontroller.py
    runs its stand-in scripts."""
    import json
    import os
    import subprocess
    import sys
    from snout_amd.message import BtleMessage
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    out = subprocess.run([sys.executable, "-m", "snout_amd.cli", "btle-rx", "-c", "37", "-g", "6",
                          "-a", "8e89bed6", "-k", "555555", "--iq",
                          os.path.join(gold, "btle_ch37_4msps.cf32")],
                         cwd=root, capture_output=True, timeout=300, check=True)
    lines = out.stdout.splitlines(keepends=True)
    truth = json.load(open(os.path.join(gold, "btle_ch37_truth.json")))
    msgs = [m for m in (BtleMessage.fromraw(ln) for ln in lines) if m]
    assert len(msgs) == 8
    for m, t in zip(msgs, truth):
        pdu = bytes.fromhex(t["pdu"])
        assert m.sender == pdu[2:8][::-1].hex() and m.payload_hex == pdu[8:].hex()
        assert m.channel == "37" and m.access_address == "8e89bed6"


@pytest.mark.parametrize("channel,aa,crc_init", [(0, 0x8E89BED6, 0x555555), (38, 0x8E89BED6, 0x555555),
                                                   (17, 0x50655D2A, 0x17B3C5), (39, 0xFFFFFFFE, 0x000001)])
def test_other_channels_and_access_addresses(oracle, channel, aa, crc_init):
    """`-c`, `-a`, `-k` of the reference's btle_rx call (snout/util/btle.py:63-68) all reach the
    kernels: whitening seed, correlator word, CRC preset."""
    from snout_amd.rx import SnoutRx
    rng = np.random.default_rng(channel)
    n = 1 << 18
    x = np.zeros(n, dtype=np.complex64)
    sent = []
    pos = 3000
    while pos + 2500 < n:
        pdu = synth.btle_random_pdu(rng)
        wave = synth.gfsk_modulate(synth.btle_air_bits(pdu, channel, aa=aa, crc_init=crc_init))
        x[pos:pos + wave.size] += wave
        sent.append(pdu)
        pos += wave.size + int(rng.integers(500, 9000))
    x += (0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    with SnoutRx(proto=0, channel=channel, access_addr=aa, crc_init=crc_init) as rx:
        got = rx.process(x)
    want, _ = oracle.btle_segment(x, channel=channel, aa=aa, crc_init=crc_init)
    _same_packets(got, want)
    assert all(p["channel"] == channel for p in got)
    if aa == 0xFFFFFFFE:
        return      # not a legal access address (31 equal bits): it matches one bit early on its own
                    # preamble, so only GPU == oracle is claimed, not that every packet decodes
    ok = [bytes(p["bytes"][:p["len"] - 3]) for p in got if p["crc_ok"]]
    assert ok == sent
    # the same capture read with the wrong whitening seed / CRC preset yields no good packet
    with SnoutRx(proto=0, channel=(channel + 1) % 40, access_addr=aa, crc_init=crc_init) as rx:
        assert not any(p["crc_ok"] for p in rx.process(x))
