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
        SnoutRx(proto=1, n_channels=16, batch_segments=9)
    plain = SnoutRx(proto=1, n_channels=16)
    with pytest.raises(SnoutError):
        plain.submit_batch([x, x], [0, 0])


def test_sharded_scan_with_batches(capture):
    from snout_amd.sharded import ShardedScan
    n_in = capture.numel() // 2
    src = lambda a, b: capture[2 * a:2 * b]
    ref = ShardedScan(1, n_channels=16, seg_len=1 << 20).run(n_in, src)
    for batch, handles in ((4, 1), (3, 2), (8, 1)):
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
