"""The C oracle run as ONE lane (core >= n: the reference's sequential receiver) against a second,
independently written statement of SURVEY.md Appendix A.2.1-A.2.4 (tests/zb_sequential_ref.py, plain
numpy / Python): soft taps and decoded records must agree record for record."""
import numpy as np
import pytest

from snout_amd import synth
import zb_sequential_ref as ref


def _records(oracle, x, channel):
    want = oracle.zigbee_segment(x, channel=channel, core=1 << 20, warmup=512)
    assert np.all(want["aux"] == 0)                    # one lane
    return want


@pytest.mark.parametrize("seed,gap,sigma,n", [(21, 3000.0, 0.05, 1 << 17), (22, 1500.0, 0.12, 1 << 17)])
def test_one_lane_oracle_equals_the_independent_sequential_receiver(oracle, seed, gap, sigma, n):
    x, truth = synth.zigbee_capture(n, channel=15, seed=seed, mean_gap=gap, sigma=sigma, max_len=60)
    assert len(truth) >= 10
    got, d, z, chips = ref.receive(x, oracle.zb_mmse_taps(), oracle.zb_chip_map(), channel=15)
    want = _records(oracle, x, 15)
    # soft taps of the C oracle (lane 0 of a one-lane run) against the numpy chain
    assert np.array_equal(oracle.zb_discrim(x), d)
    zo, co = oracle.zigbee_lane_soft(x, lane=0, core=1 << 20, warmup=512, cap=1 << 18)
    assert co.size == chips.size and np.array_equal(co, chips)
    used = z.size - 8                                  # the oracle's tap holds what its loop consumed
    assert np.array_equal(zo[:used], z[:used])
    # records: every field
    assert len(got) == len(want) >= len(truth) - 1
    for g, w in zip(got, want):
        assert g["sample_index"] == int(w["sample_index"]) and g["len"] == int(w["len"])
        assert g["lqi"] == int(w["lqi"]) and g["crc_ok"] == int(w["crc_ok"])
        assert g["bytes"] == bytes(w["bytes"][:w["len"]])
    sent = {t.payload for t in truth}
    assert sum(1 for g in got if g["crc_ok"] and g["bytes"] in sent) >= len(truth) - 1


def test_sequential_receiver_on_noise_and_on_a_truncated_frame(oracle):
    """No frames in noise; a frame cut off by the end of the capture is not reported; a PHR of zero
    still publishes one byte (the sink tests the byte count after storing a byte)."""
    rng = np.random.default_rng(5)
    noise = (0.3 * (rng.standard_normal(1 << 15) + 1j * rng.standard_normal(1 << 15))).astype(np.complex64)
    got, *_ = ref.receive(noise, oracle.zb_mmse_taps(), oracle.zb_chip_map())
    assert len(got) == len(oracle.zigbee_segment(noise, core=1 << 20)) == 0
    w = synth.oqpsk_modulate(synth.zb_frame(bytes(range(40))))
    cut = np.concatenate([np.zeros(700, np.complex64), w[:w.size // 2]])
    cut = (cut + 0.02 * (rng.standard_normal(cut.size) + 1j * rng.standard_normal(cut.size))).astype(np.complex64)
    got, *_ = ref.receive(cut, oracle.zb_mmse_taps(), oracle.zb_chip_map())
    assert len(got) == len(oracle.zigbee_segment(cut, core=1 << 20)) == 0
    # PHR = 0: build the PPDU by hand (SHR + PHR 0 + one more byte on air)
    tab = synth.zb_chip_table()
    ppdu = bytes([0, 0, 0, 0, 0xA7, 0, 0x5C, 0x33])
    chips = np.concatenate([np.concatenate([tab[b & 0xF], tab[b >> 4]]) for b in ppdu]).astype(np.float64) * 2 - 1
    shape = np.array([0.0, np.sin(np.pi / 4), 1.0, np.sin(3 * np.pi / 4)])
    i_s = np.repeat(chips[0::2], 4) * np.tile(shape, chips.size // 2)
    q_s = np.concatenate([np.zeros(2), np.repeat(chips[1::2], 4) * np.tile(shape, chips.size // 2)])[:i_s.size]
    sig = np.concatenate([np.zeros(600), i_s + 1j * q_s, np.zeros(600)]).astype(np.complex64)
    sig = (sig + 0.02 * (rng.standard_normal(sig.size) + 1j * rng.standard_normal(sig.size))).astype(np.complex64)
    got, *_ = ref.receive(sig, oracle.zb_mmse_taps(), oracle.zb_chip_map())
    want = oracle.zigbee_segment(sig, core=1 << 20)
    assert len(got) == len(want) == 1 and got[0]["len"] == int(want[0]["len"]) == 1
    assert got[0]["bytes"] == bytes(want[0]["bytes"][:1]) == b"\x5c"
