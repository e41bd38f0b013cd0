"""GPU parity for the polyphase channelizer and the wideband receive paths (cfg #3 / #4 shape):
channel IQ bit-identical to the oracle's f32 specification, decoded packets bit-exact."""
import numpy as np
import pytest

from snout_amd import synth
from snout_amd._ffi import STAGE_CHAN_IQ

pytestmark = pytest.mark.gpu


def _same_packets(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("sample_index", "proto", "channel", "len", "crc_ok", "lqi", "pdu_type", "flags", "aux"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(a["bytes"], b["bytes"])


@pytest.mark.parametrize("M,proto,n", [(40, 0, 40 * 16 + 20 * 700 + 3), (40, 0, 640), (40, 0, 659),
                                        (16, 1, 16 * 16 + 8 * 1000 + 5), (16, 1, 256), (16, 1, 263)])
def test_channelizer_bit_exact(oracle, M, proto, n):
    from snout_amd.rx import SnoutRx
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    want = oracle.pfb(x, M)
    with SnoutRx(proto=proto, n_channels=M, keep_channel_iq=True) as rx:
        rx.process(x)
        for slot in (0, 1, M // 2, M - 1):
            got = rx.soft(STAGE_CHAN_IQ, slot).view(np.complex64)
            assert got.size == want.shape[1]
            assert np.array_equal(got.view(np.uint32), want[slot].view(np.uint32)), slot


@pytest.mark.parametrize("M,proto", [(40, 0), (16, 1)])
def test_channelizer_with_non_finite_samples(oracle, M, proto):
    """A NaN / Inf sample reaches exactly the outputs whose 16-tap windows hold it (the FIR is a plain chain per output);
    everything else stays bit-identical -- also across tiles, workgroup ranges and the second FFT block of a tile."""
    from snout_amd.rx import SnoutRx
    n = M * 16 + (M // 2) * 3000 + 7
    rng = np.random.default_rng(M)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for i, v in ((5, np.nan), (n // 3, np.inf), (n // 2 + 1, -np.inf), (n - 3, np.nan)):
        x[i] = v
    want = oracle.pfb(x, M)
    with SnoutRx(proto=proto, n_channels=M, keep_channel_iq=True) as rx:
        rx.process(x)
        for slot in (0, 3, M - 1):
            got = rx.soft(STAGE_CHAN_IQ, slot).view(np.complex64)
            assert np.array_equal(np.isnan(got.view(np.float32)), np.isnan(want[slot].view(np.float32)))
            ok = ~np.isnan(got.view(np.float32))
            assert np.array_equal(got.view(np.uint32)[ok], want[slot].view(np.uint32)[ok]), slot
    assert 0 < np.isnan(want[0].view(np.float32)).sum() < want[0].size        # the NaNs stayed local


_IMPL_SNIPPET = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle_py
from snout_amd.rx import SnoutRx
from snout_amd._ffi import STAGE_CHAN_IQ
M, proto, block, legacy = %(M)d, %(proto)d, %(block)r, %(legacy)r
n = M * 16 + (M // 2) * 5000 + 3
rng = np.random.default_rng(7)
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
x[n // 2] = np.nan
want = oracle_py.pfb(x, M, block_order=block, legacy_fft=legacy)
with SnoutRx(proto=proto, n_channels=M, keep_channel_iq=True) as rx:
    rx.process(x)
    for slot in (0, 1, M // 2, M - 1):
        got = rx.soft(STAGE_CHAN_IQ, slot).view(np.complex64)
        assert got.size == want.shape[1]
        nn = np.isnan(want[slot].view(np.float32))
        assert np.array_equal(np.isnan(got.view(np.float32)), nn), slot
        assert np.array_equal(got.view(np.uint32)[~nn], want[slot].view(np.uint32)[~nn]), slot
print("equal")
'''


@pytest.mark.parametrize("impl,M,proto,block,legacy", [("valu", 40, 0, False, True), ("valu", 16, 1, False, False),
                                                        ("spec12", 40, 0, False, False), ("mfma", 40, 0, True, True)])
def test_the_kept_ab_partners_of_the_channelizer_are_bit_exact_too(impl, M, proto, block, legacy):
    """In `libsnout_rx_ab.so` (`make -C snout_amd/csrc ab`: the product sources plus `pfb.hip` and `pfb_mfma.hip`)
    `SNOUT_PFB_IMPL` (read when a handle is created) selects the kernels kept beside the shipped `pfb_spec`: round 2's
    `pfb_channelize` (valu), the 12-wave layout (spec12) -- both the plain fmaf chain of `oracle_pfb` -- and the matrix-pipe
    FIR (mfma) = `oracle_pfb_block_order` (banded-Toeplitz blocks of `v_mfma_f32_16x16x4_f32`; the snippet's input carries a
    NaN, which is where the two orders differ), bit for bit.  `pfb.hip` and `pfb_mfma.hip` keep the M = 40 FFT that rounds 1-3
    specified (Cooley-Tukey 8 x 5 with twiddles: `legacy_fft`); the shipped kernel and its 12-wave layout run the prime-factor
    form that `oracle_pfb` specifies since round 4."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SNOUT_PFB_IMPL=impl, SNOUT_RX_LIB=os.path.join(root, "snout_amd", "lib", "libsnout_rx_ab.so"))
    r = subprocess.run([sys.executable, "-c", _IMPL_SNIPPET % dict(root=root, M=M, proto=proto, block=block, legacy=legacy)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("equal"), r.stderr[-2000:]


@pytest.mark.parametrize("impl", ["valu", "mfma", "VALU", "pfb"])
def test_the_product_library_carries_one_channelizer(impl):
    """The product library refuses to pretend: any `SNOUT_PFB_IMPL` other than `spec` -- an A/B kernel it does not carry, or
    a typo -- fails the creation of a wideband handle instead of silently timing the shipped kernel under another name;
    the A/B library refuses unknown names the same way."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom snout_amd.rx import SnoutRx\nfrom snout_amd._ffi import SnoutError\n"
            "try:\n    SnoutRx(proto=0, n_channels=40)\n    print('created')\nexcept SnoutError as e:\n    print('refused', e.code)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, SNOUT_PFB_IMPL=impl))
    assert r.returncode == 0 and r.stdout.strip() == "refused -1", r.stdout + r.stderr[-2000:]
    if impl in ("VALU", "pfb"):
        env = dict(os.environ, SNOUT_PFB_IMPL=impl, SNOUT_RX_LIB=os.path.join(root, "snout_amd", "lib", "libsnout_rx_ab.so"))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and r.stdout.strip() == "refused -1", r.stdout + r.stderr[-2000:]


def test_channelizer_shorter_than_prototype():
    from snout_amd.rx import SnoutRx
    with SnoutRx(proto=0, n_channels=40) as rx:
        assert len(rx.process(np.ones(639, dtype=np.complex64))) == 0
        assert len(rx.process(np.ones(0, dtype=np.complex64))) == 0


def test_wideband_btle_matches_oracle(oracle):
    from snout_amd.rx import SnoutRx
    x, truth = synth.wideband_capture(0, 40 * 30000, seed=3, bins=[0, 1, 7, 19, 20, 21, 33, 39],
                                      mean_gap=5000.0)
    want = oracle.wideband_segment(x, 0, first_sample_index=777)
    for keep in (False, True):      # fused bit slicer (default) and the unfused channel-IQ path
        with SnoutRx(proto=0, n_channels=40, keep_channel_iq=keep) as rx:
            got = rx.process(x, first_sample_index=777)
            _same_packets(got, want)
            if not keep:
                # the fused planes hold exactly the oracle's hard bits of every channel
                from snout_amd._ffi import STAGE_BTLE_BITS
                y = oracle.pfb(x, 40)
                for slot in (0, 7, 20, 39):
                    bits = rx.soft(STAGE_BTLE_BITS, slot).astype(np.uint8)
                    assert np.array_equal(bits, oracle.btle_bits(y[slot]))
    ok = {(int(p["channel"]), bytes(p["bytes"][:p["len"] - 3])) for p in got if p["crc_ok"]}
    found = sum((t.channel, t.payload) in ok for t in truth)
    assert found >= 0.95 * len(truth) and len(truth) > 20
    # records are ordered by (bin, sample_index)
    chans = {int(c) for c in got["channel"]}
    assert chans >= {t.channel for t in truth}


@pytest.mark.parametrize("n", [640 + 20 * 3, 640 + 20 * 63, 640 + 20 * 64, 640 + 20 * 200 + 7])
def test_fused_btle_edges(oracle, n):
    from snout_amd.rx import SnoutRx
    from snout_amd._ffi import STAGE_BTLE_BITS
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    y = oracle.pfb(x, 40)
    with SnoutRx(proto=0, n_channels=40) as rx:
        got = rx.process(x)
        _same_packets(got, oracle.wideband_segment(x, 0))
        if y.shape[1] >= 5:
            for slot in (0, 39):
                assert np.array_equal(rx.soft(STAGE_BTLE_BITS, slot).astype(np.uint8), oracle.btle_bits(y[slot]))


def test_wideband_zigbee_matches_oracle(oracle):
    from snout_amd.rx import SnoutRx
    x, truth = synth.wideband_capture(1, 16 * 70000, seed=4, bins=[0, 3, 8, 11, 15],
                                      mean_gap=12000.0, max_len=50)
    with SnoutRx(proto=1, n_channels=16) as rx:
        got = rx.process(x)
    want = oracle.wideband_segment(x, 1)
    _same_packets(got, want)
    ok = {(int(p["channel"]), bytes(p["bytes"][:p["len"]])) for p in got if p["crc_ok"]}
    found = sum((t.channel, t.payload) in ok for t in truth)
    assert found >= 0.9 * len(truth) and len(truth) > 10


@pytest.mark.parametrize("n_out", [8, 9, 63, 127, 128, 129, 255, 256, 257, 128 * 5 + 31, 4096 + 700])
def test_fused_zigbee_edges(oracle, n_out):
    """Fused 802.15.4 channelizer around its tile (128 outputs), sub-block (64) and lane boundaries:
    packets and the discriminator rows it writes equal the oracle's (channelize, then discriminate)."""
    from snout_amd.rx import SnoutRx
    from snout_amd._ffi import STAGE_ZB_DISCRIM
    n = 256 + 8 * (n_out - 1) + 3
    rng = np.random.default_rng(n_out)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    if n_out > 1000:        # some structure so that the sinks have something to find
        xs, _ = synth.wideband_capture(1, n, seed=n_out, bins=[1, 9], mean_gap=3000.0, max_len=20)
        x = xs[:n]
    y = oracle.pfb(x, 16)
    assert y.shape[1] == n_out
    with SnoutRx(proto=1, n_channels=16) as rx:
        got = rx.process(x, first_sample_index=77)
        _same_packets(got, oracle.wideband_segment(x, 1, first_sample_index=77))
        if n_out >= 9:
            for slot in (0, 7, 15):
                d = rx.soft(STAGE_ZB_DISCRIM, slot)
                want = oracle.zb_discrim(y[slot])
                assert d.size == want.size == n_out
                assert np.array_equal(d.view(np.uint32), want.view(np.uint32)), slot


def test_sharded_scan_equals_one_shot(oracle):
    """cfg #5 mechanics on one GPU: overlapping segments + dedup reproduce the one-shot result."""
    import torch
    from snout_amd.rx import SnoutRx
    from snout_amd.sharded import ShardedScan
    # BTLE wideband
    x, truth = synth.wideband_capture(0, 40 * 120000, seed=7, bins=[2, 11, 20, 31], mean_gap=5000.0)
    t = torch.from_numpy(x.view(np.float32)).cuda()
    with SnoutRx(proto=0, n_channels=40) as rx:
        whole = rx.process(t)
    sc = ShardedScan(proto=0, n_channels=40, seg_len=40 * 30000)
    got = sc.run(len(x), lambda a, b: t[2 * a:2 * b])
    sc.close()
    ok_w = {(int(p["channel"]), int(p["sample_index"]), bytes(p["bytes"][:p["len"]])) for p in whole if p["crc_ok"]}
    ok_s = {(int(p["channel"]), int(p["sample_index"]), bytes(p["bytes"][:p["len"]])) for p in got if p["crc_ok"]}
    assert ok_s == ok_w and len(ok_w) >= 0.95 * len(truth)
    assert len(got) == len({(int(p["channel"]), int(p["sample_index"])) for p in got})
    # Zigbee narrowband, long frames straddling the cuts
    z, ztruth = synth.zigbee_capture(1 << 20, channel=12, seed=9, mean_gap=9000.0)
    tz = torch.from_numpy(z.view(np.float32)).cuda()
    sc = ShardedScan(proto=1, channel=12, seg_len=1 << 17)
    gz = sc.run(len(z), lambda a, b: tz[2 * a:2 * b])
    sc.close()
    good = [bytes(p["bytes"][:p["len"]]) for p in gz if p["crc_ok"]]
    sent = [t.payload for t in ztruth]
    assert len(good) == len(set(good))                        # no duplicate survives the dedup
    assert sum(s in set(good) for s in sent) >= 0.97 * len(sent)
    # a large carrier offset: the DC estimate needs ~3 time constants, so every segment but the
    # first starts 25 000 samples early and frames right behind a cut are still decoded
    z, ztruth = synth.zigbee_capture(1 << 20, channel=12, seed=10, mean_gap=9000.0, cfo_max_hz=120e3)
    tz = torch.from_numpy(z.view(np.float32)).cuda()
    sc = ShardedScan(proto=1, channel=12, seg_len=1 << 17)
    gz = sc.run(len(z), lambda a, b: tz[2 * a:2 * b])
    sc.close()
    good = [bytes(p["bytes"][:p["len"]]) for p in gz if p["crc_ok"]]
    with SnoutRx(proto=1, channel=12) as rx:
        one = {bytes(p["bytes"][:p["len"]]) for p in rx.process(tz) if p["crc_ok"]}
    sent = [t.payload for t in ztruth]
    assert len(good) == len(set(good))
    # the CFO also jumps from frame to frame here, which costs the one-shot receiver a frame too
    assert sum(s in set(good) for s in sent) >= sum(s in one for s in sent) - 1 >= 0.9 * len(sent)


def test_a_capture_that_fits_one_segment_is_not_padded(oracle):
    """a sharded scan pads a short segment to the common length only when the capture has MORE than one segment
    (so that a rank's segments are one batch).  A capture that fits one segment goes to the library at its own length: the
    same records as one process() call on it, also with a frame cut by the capture's end (padded with zeros, that frame
    was pushed on through the zeros), and no full segment's worth of channelizer and lane work for a few samples."""
    import torch
    from snout_amd.rx import SnoutRx
    from snout_amd.sharded import ShardedScan
    x, truth = synth.wideband_capture(1, 16 * 60000, seed=21, bins=[1, 6, 9, 14], mean_gap=6000.0)
    x = x[:16 * 52000 + 3200]                                       # the cut falls into frames
    t = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()
    for batch in (1, 4):
        sc = ShardedScan(proto=1, n_channels=16, seg_len=1 << 24, batch=batch)
        assert len(x) <= sc.pad_to
        sc.start(len(x), lambda a, b: t[2 * a:2 * b])
        assert [j["pad_to"] for j in sc._jobs] == [0] and sc._jobs[0]["segs"] == [(0, len(x))]
        while sc.active():
            sc.step()
        got = sc.finish(sc._parts)
        sc.close()
        from snout_amd import dist as sdist
        want = sdist.dedup_records(oracle.wideband_segment(x, 1), tol=8 * 64 + 8)     # (finish() sorts by channel number)
        assert len(got) == len(want) > 20 and got.tobytes() == want.tobytes()
    # a capture of several segments still has ONE length per submission (its last segment padded)
    sc = ShardedScan(proto=1, n_channels=16, seg_len=16 * 16384, batch=4)
    sc.start(len(x), lambda a, b: t[2 * a:2 * b])
    assert all(j["pad_to"] == sc.pad_to for j in sc._jobs) and len(sc._segs) > 2
    while sc.active():
        sc.step()
    sc.close()


def test_lanes_against_one_lane_on_a_noisy_wideband_capture():
    """The residual of the lane decomposition where it is largest (DESIGN.md deviation 3): a 16-channel
    capture with noise added on top of the channelizer's leakage, many frames with marginal chips.  One lane
    (core >= n) is the sequential receiver; the default lanes may decide marginal frames differently -- the
    timing loop on noise is not contractive, so no lane can reproduce the sequential loop's state when a
    frame arrives -- but only those: every difference is a whole frame (same bytes, never a corrupted
    one), bad-FCS records agree, and the count of differing frames stays within 2 %."""
    import collections
    import torch
    from snout_amd.rx import SnoutRx
    tz, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
    t = torch.from_numpy(np.ascontiguousarray(tz).view(np.float32)).cuda()
    torch.manual_seed(5)
    x = t.repeat(16)
    x += 0.05 * torch.randn_like(x)

    flagged = {}

    def frames(core):
        r = SnoutRx(proto=1, n_channels=16, zb_core=core).process(x)
        ok = r[r["crc_ok"] == 1]
        flagged[core] = (int(((r["flags"] & 4) != 0).sum()), len(r))
        key = lambda a: [(int(c), bytes(b[:l]), int(s)) for c, s, l, b in zip(a["channel"], a["sample_index"], a["len"], a["bytes"])]
        return key(ok), key(r[r["crc_ok"] == 0])

    def missing(A, B):
        d = collections.defaultdict(list)
        for c, b, s in B:
            d[(c, b)].append(s)
        return sum(1 for c, b, s in A if not any(abs(s - u) <= 8 for u in d.get((c, b), [])))

    one_ok, one_bad = frames(1 << 22)
    assert len(one_ok) > 900
    for core in (2048, 4096):
        ok, bad = frames(core)
        lost, extra = missing(one_ok, ok), missing(ok, one_ok)
        assert lost + extra <= 0.02 * len(one_ok), (core, lost, extra)
        assert abs(len(ok) - len(one_ok)) <= 0.01 * len(one_ok)
        assert missing(one_bad, bad) + missing(bad, one_bad) <= 2
    # SNOUT_PKT_ZB_SEAM_DISAGREED: never with one lane; with lanes it marks the frames inside which two timing loops
    # handed over while disagreeing on the frame's chips -- a few per cent here, where the noise is heavy
    assert flagged[1 << 22][0] == 0
    for core in (2048, 4096):
        assert 0 < flagged[core][0] <= 0.05 * flagged[core][1], flagged


def _frame_keys(a):
    return [(int(c), bytes(b[:l]), int(s)) for c, s, l, b in zip(a["channel"], a["sample_index"], a["len"], a["bytes"])]


def _missing(A, B):
    import collections
    d = collections.defaultdict(list)
    for c, b, s in B:
        d[(c, b)].append(s)
    return sum(1 for c, b, s in A if not any(abs(s - u) <= 8 for u in d.get((c, b), [])))


def test_default_lanes_against_one_lane_on_the_benchmarks_dense_traffic():
    """the DEFAULT 802.15.4 decode (lanes of 6144 / 3072 + the frame repair) against
    ONE sequential lane per channel -- the reference's receiver (Zigbee_rx/top_block.py:67,69) -- on the traffic cfg #4 / #5
    are timed on, built as bench.py builds it: all 16 bins busy (a transmitting neighbour 2 MHz either side of every
    channel), slotted, AWGN sigma 0.05, the bench's own 32 independently seeded tiles (2^26 input samples: ~3 500 DISTINCT
    frames, not one tile's few dozen repeated).  Round 4's lanes lost 4-7 % of the sequential receiver's frames here
    (profiles/r4_lane_residual.md).  The bound is the one the measurements defend (profiles/r6_fidelity.md, these two tile
    sets, 6 998 frames: 0.30 % lost + 0.74 % extra; with round 5's warm-up of 1024: 0.64 % + 1.10 %): lost <= 0.8 %,
    lost + extra <= 2 %.  Every
    difference is a whole frame with the bytes that were sent.  Without the repair (SNOUT_ZB_REPAIR=0 is an A/B switch of
    the library) the same shape loses several per cent."""
    import os
    import sys
    import torch
    from snout_amd.rx import SnoutRx
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda", 0)
    lost = extra = n_one = repaired = 0
    distinct = set()
    for seed in (2, 4):                                              # the tile sets of bench.py's cfg #4 and cfg #5 captures
        tiles, truths = bench.make_tiles("cfg4", seed, dev, k=16)
        sent = {t.payload for tr in truths for t in tr}
        xd = bench.resident_capture(tiles, 16 * (tiles.shape[1] // 2), seed=seed, device=dev)
        del tiles
        with SnoutRx(proto=1, n_channels=16, zb_core=1 << 22) as rx:
            one = rx.process(xd).copy()
        with SnoutRx(proto=1, n_channels=16) as rx:
            got = rx.process(xd).copy()
        del xd
        one_ok, got_ok = one[one["crc_ok"] == 1], got[got["crc_ok"] == 1]
        assert ((one["flags"] & 12) == 0).all()                      # one lane: no seams, nothing repaired
        # FCS-ok frames carry what was sent (the PSDU, FCS included), repaired ones too
        assert all(bytes(p["bytes"][:p["len"]]) in sent for p in got_ok)
        n_one += len(one_ok)
        distinct |= {(seed, c, b) for c, b, _ in _frame_keys(one_ok)}
        lost += _missing(_frame_keys(one_ok), _frame_keys(got_ok))
        extra += _missing(_frame_keys(got_ok), _frame_keys(one_ok))
        repaired += int(((got["flags"] & 8) != 0).sum())
    assert n_one > 3000 and len(distinct) > 3000 and repaired > 60
    assert lost <= 0.008 * n_one and lost + extra <= 0.02 * n_one, (lost, extra, n_one)


def test_default_lanes_against_one_lane_on_sparse_traffic():
    """VERDICT r4 item 1, the other capture: 8 of the 16 bins busy (no transmitting neighbours), frames up to 100 bytes,
    AWGN sigma 0.05, two segments of 2^24 input samples: the default decode loses <= 1 % of the sequential receiver's
    frames here too (oracle, four segments: 3 lost + 2 extra of 1 947, profiles/r5_lane_fidelity.md)."""
    import torch
    from snout_amd.rx import SnoutRx
    tile, truth = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
    sent = {t.payload for t in truth}
    lost = extra = n_one = 0
    for seed in (100, 101):
        rng = np.random.default_rng(seed)
        x = np.tile(tile, 8)
        x = (x + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        xd = torch.from_numpy(x.view(np.float32)).cuda()
        with SnoutRx(proto=1, n_channels=16, zb_core=1 << 22) as rx:
            one = rx.process(xd)
        with SnoutRx(proto=1, n_channels=16) as rx:
            got = rx.process(xd)
        one_ok, got_ok = one[one["crc_ok"] == 1], got[got["crc_ok"] == 1]
        assert all(bytes(p["bytes"][:p["len"]]) in sent for p in got_ok)
        n_one += len(one_ok)
        lost += _missing(_frame_keys(one_ok), _frame_keys(got_ok))
        extra += _missing(_frame_keys(got_ok), _frame_keys(one_ok))
    assert n_one > 800
    assert lost <= 0.01 * n_one and lost + extra <= 0.01 * n_one, (lost, extra, n_one)


@pytest.mark.parametrize("cfo_hz,sigma", [(0.0, 0.0), (50e3, 0.02), (100e3, 0.02), (50e3, 0.1)])
def test_one_clean_802154_channel_through_the_16_channel_prototype(cfo_hz, sigma):
    """the M = 16 prototype's 0.9 MHz cutoff was chosen on the synthetic all-bins raster.  Loopback on
    something else: ONE 2 Mchip/s O-QPSK channel alone in the band (no neighbours), with carrier offsets up to
    +-100 kHz and noise: the wideband path must find what the narrowband receiver finds on the same 4 Msps stream
    before it was put on the bin (the filter costs the wanted signal nothing it needs)."""
    from snout_amd.rx import SnoutRx
    b, seed, n_ch = 5, 21, 1 << 20
    x, truth = synth.wideband_capture(1, 8 * n_ch, seed=seed, bins=[b], sigma=sigma, cfo_max_hz=cfo_hz, slotted=False)
    nb, tr = synth.zigbee_capture(n_ch, channel=synth.zigbee_bin_channel(b), seed=seed * 1000 + b, noise=False,
                                  cfo_max_hz=cfo_hz, slot_phase=None)
    assert [t.payload for t in tr] == [t.payload for t in truth] and len(truth) > 30
    with SnoutRx(proto=1, n_channels=16) as rx:
        wide = rx.process(x)
    with SnoutRx(proto=1, channel=synth.zigbee_bin_channel(b)) as rx:
        narrow = rx.process(nb)                                       # the clean narrowband stream: what there is to find
    sent = {t.payload for t in truth}
    ok_w = {bytes(p["bytes"][:p["len"]]) for p in wide if p["crc_ok"] and p["channel"] == synth.zigbee_bin_channel(b)}
    ok_n = {bytes(p["bytes"][:p["len"]]) for p in narrow if p["crc_ok"]}
    assert ok_w <= sent and ok_n <= sent
    print("loopback", cfo_hz, sigma, "sent", len(sent), "narrowband", len(ok_n), "wideband", len(ok_w))
    assert len(ok_n) >= 0.7 * len(sent)                  # (the reference chain itself loses frames at large carrier offsets: 24 of 32 at 100 kHz)
    assert len(ok_w) >= (0.95 if sigma <= 0.02 else 0.9) * len(ok_n), (len(ok_w), len(ok_n), len(sent))
    assert not [p for p in wide if p["crc_ok"] and p["channel"] != synth.zigbee_bin_channel(b)]     # nothing leaks into other bins
