"""GPU: the pipelined form of the C ABI under awkward use — segment sizes that grow and shrink
while three are in flight (buffers of every work set are re-sized), all four handle kinds, a
handle destroyed with segments pending, and the device copy of the records."""
import ctypes as C

import numpy as np
import pytest

from snout_amd import synth
from snout_amd._ffi import PKT_DTYPE

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert len(a) == len(b)
    for f in ("sample_index", "proto", "channel", "len", "crc_ok", "lqi", "pdu_type", "flags", "aux"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(a["bytes"], b["bytes"])


def _captures(kind, sizes):
    out = []
    for i, n in enumerate(sizes):
        if kind == "btle":
            x, _ = synth.btle_capture(n, seed=60 + i, mean_gap=3000.0)
        elif kind == "zigbee":
            x, _ = synth.zigbee_capture(n, seed=60 + i, mean_gap=5000.0)
        elif kind == "btle40":
            x, _ = synth.wideband_capture(0, n, seed=60 + i, bins=[3, 17, 33], mean_gap=4000.0)
        else:
            x, _ = synth.wideband_capture(1, n, seed=60 + i, bins=[2, 9], mean_gap=5000.0, max_len=40)
        out.append(x[:n])
    return out


@pytest.mark.parametrize("kind,kw,sizes", [
    ("btle", dict(proto=0, channel=37), [1 << 16, 1 << 19, 3000, 1 << 18, 77, 1 << 19, 1 << 14]),
    ("zigbee", dict(proto=1, channel=11), [1 << 16, 1 << 19, 3000, 1 << 18, 77, 1 << 19, 1 << 14]),
    ("btle40", dict(proto=0, n_channels=40), [40 * 3000, 40 * 20000, 700, 40 * 9000, 40 * 20000]),
    ("zigbee16", dict(proto=1, n_channels=16), [16 * 9000, 16 * 40000, 300, 16 * 20000, 16 * 40000]),
])
def test_sizes_change_while_segments_are_in_flight(kind, kw, sizes):
    import torch
    from snout_amd.rx import SnoutRx
    caps = _captures(kind, sizes)
    dev = [torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda() for x in caps]
    with SnoutRx(**kw) as ref:
        want = [ref.process(t, first_sample_index=1000 * i) for i, t in enumerate(dev)]
    with SnoutRx(**kw) as rx:
        got = []
        for i, t in enumerate(dev):
            rx.submit(t, first_sample_index=1000 * i)
            if i >= 2:
                got.append(rx.collect())
        got += [rx.collect(), rx.collect()]
        for a, b in zip(got, want):
            _same(a, b)
        # the device copy of the last collected records is the same data
        ptr, n = rx.last_records_device()
        assert n == len(got[-1])
        if n:
            from snout_amd.dist import _device_bytes
            raw = _device_bytes(torch, ptr, n * PKT_DTYPE.itemsize, t.device).cpu().numpy().view(PKT_DTYPE)
            _same(raw, got[-1])


def test_destroy_with_segments_pending():
    import torch
    from snout_amd.rx import SnoutRx
    x, _ = synth.zigbee_capture(1 << 19, seed=5)
    t = torch.from_numpy(x.view(np.float32)).cuda()
    for kw in (dict(proto=1, channel=11), dict(proto=0, channel=37)):
        rx = SnoutRx(**kw)
        rx.submit(t); rx.submit(t); rx.submit(t)
        rx.close()                      # waits for the device, frees everything
    torch.cuda.synchronize()
