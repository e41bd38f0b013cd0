"""GPU: several equal-length capture segments as ONE submission (`snout_rx_submit_batch_dev`) give exactly
the records of one submission per segment -- same bytes, same order (segment, channel, sample_index) --
and the sharded scan built on it returns what the one-segment-at-a-time scan returns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capture():
    import torch
    from snout_amd import synth
    x, _ = synth.wideband_capture(1, 16 * (1 << 19), seed=21, sigma=0.02)      # 2^23 input samples, 16 channels
    return torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()


def _segments(n_in, seg, overlap):
    out, a = [], 0
    while a + seg <= n_in:
        out.append((a, a + seg))
        a += seg - overlap
    return out


def test_batch_equals_single_submissions(capture):
    from snout_amd._ffi import PKT_DTYPE
    from snout_amd.rx import SnoutRx
    n_in = capture.numel() // 2
    seg, ov = 1 << 21, 1 << 18
    segs = _segments(n_in, seg, ov)[:4]
    assert len(segs) == 4
    firsts = [a // 8 for a, _ in segs]
    mins = [0] + [f + (ov // 8) // 2 for f in firsts[1:]]        # each segment leaves half its overlap to the one before
    xs = [capture[2 * a:2 * b] for a, b in segs]
    one = SnoutRx(proto=1, n_channels=16)
    per_seg = []
    for x, f, m in zip(xs, firsts, mins):
        r = one.process(x, first_sample_index=f)
        per_seg.append(r[r["sample_index"] >= m])
    assert sum(len(r) for r in per_seg) > 40 and sum(int((r["crc_ok"] == 1).sum()) for r in per_seg) > 30
    assert any((one.process(x, first_sample_index=f)["sample_index"] < m).any() for x, f, m in zip(xs[1:], firsts[1:], mins[1:]))
    rx = SnoutRx(proto=1, n_channels=16, batch_segments=4)
    for count in (4, 3, 1):
        rx.submit_batch(xs[:count], firsts[:count], mins[:count])
        got = rx.collect()
        ref = np.concatenate(per_seg[:count])
        assert got.dtype == PKT_DTYPE and len(got) == len(ref)
        assert got.tobytes() == ref.tobytes()
    # without min indices nothing is dropped; plain submit still works on a batch handle
    rx.submit_batch(xs[:2], firsts[:2])
    got = rx.collect()
    ref = np.concatenate([one.process(x, first_sample_index=f) for x, f in zip(xs[:2], firsts[:2])])
    assert got.tobytes() == ref.tobytes()
    rx.submit(xs[1], first_sample_index=firsts[1])
    assert rx.collect().tobytes() == one.process(xs[1], first_sample_index=firsts[1]).tobytes()


def test_batch_argument_checks(capture):
    from snout_amd._ffi import SnoutError
    from snout_amd.rx import SnoutRx
    x = capture[: 2 * (1 << 20)]
    rx = SnoutRx(proto=1, n_channels=16, batch_segments=2)
    with pytest.raises(SnoutError):
        rx.submit_batch([x, x, x], [0, 0, 0])                   # more than the handle was created for
    with pytest.raises(ValueError):
        rx.submit_batch([x, x[:-16]], [0, 0])                   # unequal lengths
    with pytest.raises(SnoutError):
        SnoutRx(proto=0, n_channels=1, batch_segments=2)        # wideband handles only
    with pytest.raises(SnoutError):
        SnoutRx(proto=1, n_channels=16, batch_segments=2, keep_channel_iq=True)
    with pytest.raises(SnoutError):
        SnoutRx(proto=1, n_channels=16, batch_segments=65)
    plain = SnoutRx(proto=1, n_channels=16)
    with pytest.raises(SnoutError):
        plain.submit_batch([x, x], [0, 0])


def test_sharded_scan_with_batches(capture):
    from snout_amd.sharded import ShardedScan
    n_in = capture.numel() // 2
    src = lambda a, b: capture[2 * a:2 * b]
    ref = ShardedScan(1, n_channels=16, seg_len=1 << 20).run(n_in, src)
    for batch, handles in ((4, 1), (3, 2), (8, 1), (64, 1)):
        got = ShardedScan(1, n_channels=16, seg_len=1 << 20, batch=batch, handles=handles).run(n_in, src)
        assert len(got) == len(ref) > 40
        assert got.tobytes() == ref.tobytes()


def test_btle_batch_equals_single_submissions():
    import torch
    from snout_amd import synth
    from snout_amd.rx import SnoutRx
    from snout_amd.sharded import ShardedScan
    x, _ = synth.wideband_capture(0, 40 * (1 << 17), seed=22, sigma=0.02)      # 5.2e6 input samples, 40 channels
    cap = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()
    n_in = cap.numel() // 2
    seg, ov = 40 * (1 << 15), 40 * 2048
    segs = _segments(n_in, seg, ov)[:4]
    firsts = [a // 20 for a, _ in segs]
    mins = [0] + [f + 1024 for f in firsts[1:]]
    xs = [cap[2 * a:2 * b] for a, b in segs]
    one = SnoutRx(proto=0, n_channels=40)
    per_seg = []
    for xk, f, m in zip(xs, firsts, mins):
        r = one.process(xk, first_sample_index=f)
        per_seg.append(r[r["sample_index"] >= m])
    assert sum(len(r) for r in per_seg) > 100
    rx = SnoutRx(proto=0, n_channels=40, batch_segments=4)
    for count in (4, 2, 1, 3):
        rx.submit_batch(xs[:count], firsts[:count], mins[:count])
        got = rx.collect()
        ref = np.concatenate(per_seg[:count])
        assert len(got) == len(ref) and got.tobytes() == ref.tobytes()
    src = lambda a, b: cap[2 * a:2 * b]
    ref = ShardedScan(0, n_channels=40, seg_len=seg).run(n_in, src)
    got = ShardedScan(0, n_channels=40, seg_len=seg, batch=4).run(n_in, src)
    assert len(ref) > 100 and got.tobytes() == ref.tobytes()


def test_poll_tells_when_collect_will_not_wait(capture):
    import time
    from snout_amd.rx import SnoutRx
    rx = SnoutRx(proto=1, n_channels=16)
    assert rx.ready() is False                                   # nothing pending
    x = capture[: 2 * (1 << 21)]
    rx.submit(x)
    t0 = time.time()
    while not rx.ready():
        assert time.time() - t0 < 30
        time.sleep(0.0005)
    assert len(rx.collect()) > 0 and rx.ready() is False


def test_batch_survives_a_capacity_rerun():
    """A batch whose candidates outgrow the provisioned capacity is run again as a whole with more room
    (the same rerun a single segment gets): records unchanged."""
    import torch
    from snout_amd import synth
    from snout_amd.rx import SnoutRx
    x, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=23, sigma=0.02, mean_gap=3000.0)
    cap = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()
    n = cap.numel() // 2 // 3 // 40 * 40
    xs = [cap[2 * k * n:2 * (k + 1) * n] for k in range(3)]
    firsts = [k * n // 20 for k in range(3)]
    one = SnoutRx(proto=0, n_channels=40)
    ref = np.concatenate([one.process(xk, first_sample_index=f) for xk, f in zip(xs, firsts)])
    assert len(ref) > 200
    small = SnoutRx(proto=0, n_channels=40, batch_segments=3, max_hits=64)     # far fewer than the candidates
    for _ in range(2):                                                          # the grown capacity is kept
        small.submit_batch(xs, firsts)
        got = small.collect()
        assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("proto,M,seg,count", [(0, 40, 40 * 4096, 31), (1, 16, 16 * 16384, 29), (0, 40, 40 * 1024, 64), (1, 16, 16 * 8192, 64)])
def test_large_batches_equal_single_submissions(proto, M, seg, count):
    """Up to 64 segments per submission (cfg #5 hands a rank's 48 BTLE / 20 802.15.4 segments of a step over as one):
    the channelizer cuts every segment into ranges that fill whole rounds of workgroups, the slots (segment, channel) go
    through every later kernel together -- the records are those of `count` single submissions, byte for byte."""
    import torch
    from snout_amd import synth
    from snout_amd.rx import SnoutRx
    ov = M * 512
    n_in = count * (seg - ov) + ov
    x, _ = synth.wideband_capture(proto, n_in, seed=70 + count, sigma=0.02, mean_gap=6000.0, **({} if proto == 0 else {"max_len": 30}))
    cap = torch.from_numpy(np.ascontiguousarray(x[:n_in]).view(np.float32)).cuda()
    segs = _segments(n_in, seg, ov)
    assert len(segs) == count
    D = M // 2
    firsts = [a // D for a, _ in segs]
    mins = [0] + [f + 100 for f in firsts[1:]]
    xs = [cap[2 * a:2 * b] for a, b in segs]
    one = SnoutRx(proto=proto, n_channels=M)
    per_seg = []
    for xk, f, m in zip(xs, firsts, mins):
        r = one.process(xk, first_sample_index=f)
        per_seg.append(r[r["sample_index"] >= m])
    ref = np.concatenate(per_seg)
    assert len(ref) > 50
    rx = SnoutRx(proto=proto, n_channels=M, batch_segments=count)
    rx.submit_batch(xs, firsts, mins)
    got = rx.collect()
    assert len(got) == len(ref) and got.tobytes() == ref.tobytes()


def test_records_stay_on_the_device():
    """`SNOUT_CFG_RECORDS_ON_DEVICE`: collect hands out the COUNT, nothing is downloaded; `snout_rx_pack_last_records`
    packs the device copy (from record `skip` on) and raises a device word to the longest record; `snout_rx_collect` with a
    buffer still downloads on request."""
    import ctypes as C
    import torch
    from snout_amd import _ffi, synth
    from snout_amd.rx import SnoutRx
    x, _ = synth.wideband_capture(0, 40 * (1 << 15), seed=23, sigma=0.02, mean_gap=4000.0)
    cap = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()
    with SnoutRx(proto=0, n_channels=40) as ref:
        want = ref.process(cap, first_sample_index=5)
    assert len(want) > 30
    with SnoutRx(proto=0, n_channels=40, records_on_device=True) as rx:
        rx.submit(cap, first_sample_index=5)
        n = rx.collect()
        assert isinstance(n, int) and n == len(want)
        ptr, n_dev = rx.last_records_device()
        assert n_dev == n and ptr
        W = 96
        dst = torch.zeros(n * W, dtype=torch.uint8, device="cuda")
        longest = torch.zeros(1, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        own = int(np.sort(want["sample_index"])[len(want) // 3])
        assert rx.pack_last_records(dst.data_ptr(), n, W, own, st, skip=0, longest_ptr=longest.data_ptr()) == n
        torch.cuda.synchronize()
        got = dst.cpu().numpy().view(np.uint8).reshape(n, W)
        raw = want.view(np.uint8).reshape(n, 160)[:, :W].copy()
        dropped = want["sample_index"] < own
        assert dropped.any() and not dropped.all()
        raw[dropped, :8] = np.frombuffer(np.uint64(1 << 62).tobytes(), dtype=np.uint8)
        assert np.array_equal(got, raw)
        assert int(longest.item()) == int(want["len"].max())
        # skip: the tail of the records only; a too small destination reports the overflow
        assert rx.pack_last_records(dst.data_ptr(), n, W, 0, st, skip=n - 7) == 7
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy()[:7 * W].reshape(7, W), want.view(np.uint8).reshape(n, 160)[n - 7:, :W])
        assert rx.pack_last_records(dst.data_ptr(), 3, W, 0, st, skip=n - 7) == 3
        # the copying collect still delivers records (downloaded on request)
        rx.submit(cap, first_sample_index=5)
        out = np.zeros(n + 8, dtype=_ffi.PKT_DTYPE)
        n_out = C.c_uint64(0)
        _ffi.check(rx._lib.snout_rx_collect(rx._h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n_out)))
        assert n_out.value == n and out[:n].tobytes() == want.tobytes()
        # and the synchronous entry point is unchanged
        assert rx.process(cap, first_sample_index=5).tobytes() == want.tobytes()


def test_the_default_lane_shape_is_the_same_for_a_batch_and_its_segments(oracle):
    """cfg.zb_core = cfg.zb_warmup = 0: ONE default lane shape (6144 / 3072, snout_zigbee_lane_shape) whatever a submission
    carries (round 4 chose the shape by channels x samples x segments of the submission, so the last, shorter
    batch of a capture -- or another world size -- decoded another frame set).  33 segments of 2^23 input samples in one
    batch decode what each decodes alone, which is what the oracle decodes with the default shape; an explicit zb_core
    still pins another shape for both."""
    import torch
    from snout_amd import synth
    from snout_amd.rx import SnoutRx
    n_in = 1 << 23
    x, truth = synth.wideband_capture(1, n_in, seed=91, sigma=0.05, mean_gap=9000.0)
    cap = torch.from_numpy(np.ascontiguousarray(x[:n_in]).view(np.float32)).cuda()
    count = 33          # 33 x 16 x 1 048 545 channel samples: above round 4's threshold of 2^29 (one segment is below)
    firsts = [1000 * k for k in range(count)]
    oracle.set_threads(oracle.hw_threads())
    try:
        want = oracle.wideband_segment(x[:n_in], 1)
        want_short = oracle.wideband_segment(x[:n_in], 1, core=2048, warmup=512)
    finally:
        oracle.set_threads(1)
    assert len(want) > 300 and oracle.zb_auto_shape(16) == (6144, 3072)
    with SnoutRx(proto=1, n_channels=16, batch_segments=count) as rx:
        rx.submit_batch([cap] * count, firsts)
        got = rx.collect()
        assert len(got) == count * len(want)
        for k in (0, 13, count - 1):
            seg = got[k * len(want):(k + 1) * len(want)].copy()
            seg["sample_index"] -= np.uint64(firsts[k])
            assert seg.tobytes() == want.tobytes()
        assert rx.process(cap).tobytes() == want.tobytes()          # one segment alone: the same shape, the same records
    with SnoutRx(proto=1, n_channels=16, batch_segments=count, zb_core=2048, zb_warmup=512) as rx:
        rx.submit_batch([cap] * count, firsts)
        got = rx.collect()
        assert got[:len(want_short)].tobytes() == want_short.tobytes() and len(got) == count * len(want_short)
