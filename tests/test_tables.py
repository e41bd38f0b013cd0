"""Known answers for the generated tables, stated WITHOUT the generators or the shared .inc text as the source of
truth: the product and the oracle include byte-identical copies, so an error in a table is
common-mode for every GPU-vs-oracle test.  Here the committed numbers are parsed as data and checked against what
the definitions imply: the fractional-delay bank's end rows, its mirror symmetry, unit DC gain and interpolation
property; the channelizer prototypes' symmetry, unit DC gain, cutoff and stop-band; the twiddles against exp()."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _floats(text):
    return np.array([float(x) for x in re.findall(r"[-+]?\d\.\d+e[-+]\d+", text)], dtype=np.float64)


def _mmse(path):
    return _floats(open(os.path.join(ROOT, path)).read()).reshape(129, 8)


def _pfb(path):
    out = {}
    for m in re.finditer(r"static const float (\w+)\[(\d+)\] = \{(.*?)\};", open(os.path.join(ROOT, path)).read(), flags=re.S):
        out[m.group(1)] = _floats(m.group(3))
        assert out[m.group(1)].size == int(m.group(2))
    return out


def test_both_copies_of_every_table_hold_the_same_numbers():
    assert np.array_equal(_mmse("oracle/mmse_taps.inc"), _mmse("snout_amd/csrc/mmse_taps.inc"))
    a, b = _pfb("oracle/pfb_tables.inc"), _pfb("snout_amd/csrc/pfb_tables.inc")
    assert set(a) == set(b) == {"kPfbProto40", "kPfbProto16", "kTw40", "kTw16", "kTw5"}
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_mmse_bank_known_answers():
    t = _mmse("oracle/mmse_taps.inc")
    # rows 0 and 128: unit taps on in[3] / in[4] of the 8-sample window, stored newest-first
    assert t[0].tolist() == [0, 0, 0, 0, 1, 0, 0, 0] and t[128].tolist() == [0, 0, 0, 1, 0, 0, 0, 0]
    # delay mu and delay 1 - mu are mirror images of each other
    assert np.allclose(t, t[::-1, ::-1], atol=3e-7)
    # row 64 (half a sample): symmetric, its two centre taps carry most of the weight
    assert np.allclose(t[64], t[64][::-1], atol=3e-7) and 0.60 < t[64][3] < 0.64 and abs(t[64][3] - t[64][4]) < 3e-7
    # every row passes DC unchanged and reproduces a slow sinusoid at its fractional delay
    assert np.allclose(t.sum(axis=1), 1.0, atol=2e-3)
    n = np.arange(-8, 16, dtype=np.float64)
    for f in (0.02, 0.1):                                # cycles per sample, well inside the 0.25 design band
        x = np.cos(2 * np.pi * f * n)
        for row in (16, 64, 100):
            mu = row / 128.0
            y = sum(t[row][k] * x[8 + 7 - k] for k in range(8))         # taps newest-first over in[0..7] = x[8..15]
            assert abs(y - np.cos(2 * np.pi * f * (3 + mu))) < 5e-3, (f, row)


def test_prototype_known_answers():
    tabs = _pfb("oracle/pfb_tables.inc")
    for M, cutoff_rel in ((40, 1.0), (16, 0.9)):
        h = tabs["kPfbProto%d" % M]
        L = 16 * M
        assert h.size == L and np.allclose(h, h[::-1], atol=1e-9)             # linear phase
        assert abs(h.sum() - 1.0) < 2e-6                                     # unit DC gain
        assert np.argmax(h) in (L // 2 - 1, L // 2)
        H = np.abs(np.fft.rfft(h, 1 << 16))
        f = np.arange(H.size) / float(1 << 16)                               # cycles per input sample
        spacing = 1.0 / M
        half = cutoff_rel * spacing / 2                                      # the -6 dB point of a windowed sinc
        assert abs(20 * np.log10(H[np.argmin(abs(f - half))]) + 6.0) < 0.5
        assert 20 * np.log10(H[f >= 1.25 * spacing].max()) < -60             # beyond the neighbour's centre + 25 %
        assert 20 * np.log10(H[f <= 0.2 * spacing].min()) > -0.5             # flat where the wanted signal sits


def test_twiddles_against_exp():
    tabs = _pfb("oracle/pfb_tables.inc")
    for N in (40, 16, 5):
        w = np.exp(-2j * np.pi * np.arange(N) / N)
        got = tabs["kTw%d" % N].reshape(N, 2)
        want = np.stack([w.real, w.imag], 1).astype(np.float32)
        assert np.array_equal(got.astype(np.float32), want), N       # the text carries each f32 value exactly (10 digits)
