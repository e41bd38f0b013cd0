"""CPU tests pinning the BTLE oracle (oracle/oracle_btle.c): known answers, three independent
implementations of whitening/CRC, and TX->RX loopback on the committed cfg #1 fixture."""
import json
import os

import numpy as np
import pytest

from snout_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_whitening_known_answer(oracle):
    # First bytes of the BLE channel-37 whitening sequence as widely published for
    # software-defined advertisers (8D D2 57 A1 3D A7 66 B0 ...).
    assert oracle.btle_whiten_seq(37, 8).hex() == "8dd257a13da766b0"


@pytest.mark.parametrize("ch", [0, 1, 10, 11, 36, 37, 38, 39])
def test_whitening_matches_independent_bit_serial(oracle, ch):
    want = synth.bits_to_bytes_lsb(synth.btle_whiten_bits(ch, 42 * 8))
    assert oracle.btle_whiten_seq(ch, 42) == want
    # period of a maximal 7-bit LFSR
    bits = synth.btle_whiten_bits(ch, 254)
    assert np.array_equal(bits[:127], bits[127:])


def test_crc24_three_implementations_agree(oracle):
    rng = np.random.default_rng(0)
    for n in [0, 1, 2, 8, 39]:
        for _ in range(20):
            d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            init = int(rng.integers(0, 1 << 24))
            a = oracle.btle_crc24(d, init)
            b = oracle.btle_crc24(d, init, table=True)
            c = synth.btle_crc24_bits(synth.bytes_to_bits_lsb(d), init)
            assert a == b == sum(bit << (23 - i) for i, bit in enumerate(c))


def test_crc24_is_linear_and_detects_single_bit_errors(oracle):
    rng = np.random.default_rng(1)
    x = bytes(rng.integers(0, 256, 20, dtype=np.uint8))
    y = bytes(rng.integers(0, 256, 20, dtype=np.uint8))
    z = bytes(a ^ b for a, b in zip(x, y))
    assert oracle.btle_crc24(x, 0) ^ oracle.btle_crc24(y, 0) == oracle.btle_crc24(z, 0)
    base = oracle.btle_crc24(x)
    for bit in range(0, 160, 7):
        e = bytearray(x)
        e[bit // 8] ^= 1 << (bit % 8)
        assert oracle.btle_crc24(bytes(e)) != base


def test_cfg1_fixture_decodes_8_of_8(oracle):
    """SURVEY §8d cfg #1: the committed 1 MB ch37 capture; all 8 packets CRC-ok, bytes equal to
    what the generator sent."""
    x = np.fromfile(os.path.join(GOLD, "btle_ch37_4msps.cf32"), dtype=np.complex64)
    truth = json.load(open(os.path.join(GOLD, "btle_ch37_truth.json")))
    pk, _ = oracle.btle_segment(x, channel=37)
    assert len(pk) == 8 == len(truth)
    for p, t in zip(pk, truth):
        assert p["crc_ok"] == 1
        assert bytes(p["bytes"][:p["len"] - 3]).hex() == t["pdu"]
        assert abs(int(p["sample_index"]) - t["sample_index"]) <= 3
        assert p["channel"] == 37 and p["proto"] == 0


def test_wrong_channel_whitening_fails_crc(oracle):
    x = np.fromfile(os.path.join(GOLD, "btle_ch37_4msps.cf32"), dtype=np.complex64)
    pk, _ = oracle.btle_segment(x, channel=38)
    assert not any(p["crc_ok"] for p in pk)


def test_search_resumes_after_packet(oracle):
    """Every sampling phase that matches the access address is a hit, but the sequential search
    reports one packet and resumes after its CRC."""
    x, truth = synth.btle_capture(1 << 17, seed=3, mean_gap=4000.0)
    bits = oracle.btle_bits(x)
    all_hits = oracle.btle_all_hits(bits)
    pk, examined = oracle.btle_segment(x)
    assert len(all_hits) > len(pk) == len(truth)
    assert set(examined.tolist()) <= set(all_hits.tolist())
    ends = pk["sample_index"] + 128 + 32 * pk["len"].astype(np.uint64)
    assert np.all(pk["sample_index"][1:] >= ends[:-1])


def test_bits_definition_and_edges(oracle):
    rng = np.random.default_rng(2)
    for n in [0, 1, 4, 5, 6, 100]:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        b = oracle.btle_bits(x)
        assert b.size == max(0, n - 4)
        if n > 4:
            a = x.real[:-4] * x.imag[4:]
            c = x.real[4:] * x.imag[:-4]
            assert np.array_equal(b, (a > c).astype(np.uint8))


def test_truncated_packet_is_skipped_not_reported(oracle):
    x, truth = synth.btle_capture(1 << 16, seed=9, mean_gap=3000.0, sigma=0.01)
    last = truth[-1]
    cut = last.sample_index + 128 + 32 * 4          # header complete, payload cut
    pk, _ = oracle.btle_segment(x[:cut])
    assert len(pk) == len(truth) - 1
    pk, _ = oracle.btle_segment(x[:last.sample_index + 100])   # access address itself cut
    assert len(pk) == len(truth) - 1


def test_gfsk_modulator_is_constant_envelope_and_signed():
    bits = np.array([1, 1, 1, 1, 0, 0, 0, 0, 1, 0, 1, 0], dtype=np.uint8)
    w = synth.gfsk_modulate(bits)
    assert np.allclose(np.abs(w), 1.0, atol=1e-5)
    d = np.angle(w[4:] * np.conj(w[:-4]))
    mid = lambda k: d[(4 + k) * 4 + 0]     # noqa: E731  pad 4 symbols, symbol k centre-ish
    assert mid(1) > 0 and mid(2) > 0 and mid(5) < 0 and mid(6) < 0
