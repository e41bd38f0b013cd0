"""Deterministic synthetic IQ generators (numpy) for the receive path.

These are the *signal definitions* the receive path is measured on:

* BTLE advertising packets -> 1 Mbit/s GFSK (BT 0.5, h 0.5) at 4 samples/symbol, the rate
  ``btle_rx`` runs at (reference call site ``snout/util/btle.py:63-69``; access address and CRC
  init are the ``-a 8e89bed6 -k 555555`` of that call).
* IEEE 802.15.4 O-QPSK frames built the way the reference's own transmitter flowgraph does it
  (``snout/modulations/Zigbee/hackrf/Zigbee_tx/top_block.py:59-71``): nibble -> 16 complex chips
  (even chip on I, odd chip on Q) -> repeat 4 -> half-sine ``[0, sin(pi/4), 1, sin(3pi/4)]`` ->
  Q delayed by 2 samples, i.e. 4 samples per complex chip pair = 2 samples per chip at 4 Msps.
* A wideband compositor that places narrowband channels on the bin centres of the polyphase
  channelizer.

Whitening / CRC here are written independently of ``oracle/`` (bit-serial, straight from the
Bluetooth Core Spec Vol 6 Part B §3.1-3.2) so tests can cross-check three implementations.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

BTLE_ADV_AA = 0x8E89BED6
BTLE_ADV_CRC_INIT = 0x555555
BTLE_SPS = 4

ADV_PDU_TYPES = ["ADV_IND", "ADV_DIRECT_IND", "ADV_NONCONN_IND", "SCAN_REQ",
                 "SCAN_RSP", "CONNECT_REQ", "ADV_SCAN_IND"]


# ------------------------------------------------------------------------------------------------
# BTLE bit-level
# ------------------------------------------------------------------------------------------------
def btle_whiten_bits(channel: int, nbits: int) -> np.ndarray:
    """Whitening bit sequence: LFSR x^7+x^4+1, position 0 = 1, positions 1..6 = channel MSB..LSB."""
    reg = [1] + [(channel >> (5 - i)) & 1 for i in range(6)]
    out = np.empty(nbits, dtype=np.uint8)
    for i in range(nbits):
        o = reg[6]
        out[i] = o
        reg = [o, reg[0], reg[1], reg[2], reg[3] ^ o, reg[4], reg[5]]
    return out


def bytes_to_bits_lsb(data: bytes) -> np.ndarray:
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    return np.unpackbits(a, bitorder="little")


def bits_to_bytes_lsb(bits: np.ndarray) -> bytes:
    return np.packbits(np.asarray(bits, dtype=np.uint8), bitorder="little").tobytes()


def btle_crc24_bits(bits: Sequence[int], init: int = BTLE_ADV_CRC_INIT) -> List[int]:
    """CRC24 (x^24+x^10+x^9+x^6+x^4+x^3+x+1) over a bit sequence; returns the 24 CRC bits in
    transmit order (register position 23 first)."""
    r = init & 0xFFFFFF
    for b in bits:
        t = (r >> 23) & 1
        r = (r << 1) & 0xFFFFFF
        if t != int(b):
            r ^= 0x00065B
    return [(r >> (23 - i)) & 1 for i in range(24)]


def btle_adv_pdu(pdu_type: int, adva: bytes, advdata: bytes = b"", txadd: int = 0,
                 rxadd: int = 0) -> bytes:
    """Advertising-channel PDU (header + payload). ``adva`` is given MSB first (as btle_rx
    prints it, message.py:214) and is sent LSB first."""
    assert len(adva) == 6
    payload = bytes(adva[::-1]) + bytes(advdata)
    assert 6 <= len(payload) <= 37
    h0 = (pdu_type & 0x0F) | ((txadd & 1) << 6) | ((rxadd & 1) << 7)
    return bytes([h0, len(payload)]) + payload


def btle_air_bits(pdu: bytes, channel: int, aa: int = BTLE_ADV_AA,
                  crc_init: int = BTLE_ADV_CRC_INIT) -> np.ndarray:
    """Preamble + access address + whitened(PDU + CRC), in transmit order."""
    pdu_bits = bytes_to_bits_lsb(pdu)
    crc_bits = np.array(btle_crc24_bits(pdu_bits, crc_init), dtype=np.uint8)
    body = np.concatenate([pdu_bits, crc_bits])
    body ^= btle_whiten_bits(channel, body.size)
    aa_bits = np.array([(aa >> i) & 1 for i in range(32)], dtype=np.uint8)
    # alternating preamble whose first bit equals the first (LSB) bit of the access address
    first = aa_bits[0]
    pre = np.array([first ^ (i & 1) for i in range(8)], dtype=np.uint8)
    return np.concatenate([pre, aa_bits, body])


# ------------------------------------------------------------------------------------------------
# GFSK modulator
# ------------------------------------------------------------------------------------------------
def _gauss_taps(bt: float, sps: int, span: int) -> np.ndarray:
    t = (np.arange(span * sps + 1) - span * sps / 2.0) / sps
    sigma = math.sqrt(math.log(2.0)) / (2.0 * math.pi * bt)
    g = np.exp(-0.5 * (t / sigma) ** 2)
    return g / g.sum()


def gfsk_modulate(bits: np.ndarray, sps: int = BTLE_SPS, bt: float = 0.5, h: float = 0.5,
                  pad_symbols: int = 4) -> np.ndarray:
    """bits -> complex64 baseband, constant envelope. Bit 1 = positive frequency deviation."""
    nrz = 2.0 * np.asarray(bits, dtype=np.float64) - 1.0
    nrz = np.concatenate([np.zeros(pad_symbols), nrz, np.zeros(pad_symbols)])
    up = np.repeat(nrz, sps)
    f = np.convolve(up, _gauss_taps(bt, sps, 4), mode="same")
    phase = np.cumsum(f) * (math.pi * h / sps)
    return np.exp(1j * phase).astype(np.complex64)


@dataclass
class TruthPacket:
    proto: int
    channel: int
    sample_index: int      # first sample of the access address (BTLE) / of the preamble (Zigbee)
    payload: bytes         # BTLE: PDU (header+payload); Zigbee: PSDU incl. FCS
    extra: dict = field(default_factory=dict)


def btle_random_pdu(rng: np.random.Generator, max_advdata: int = 31) -> bytes:
    pdu_type = int(rng.choice([0, 2]))              # ADV_IND / ADV_NONCONN_IND
    adva = bytes(rng.integers(0, 256, 6, dtype=np.uint8))
    adva = bytes([adva[0] | 0xC0]) + adva[1:]       # random static address
    n = int(rng.integers(0, max_advdata + 1))
    advdata = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    return btle_adv_pdu(pdu_type, adva, advdata, txadd=1, rxadd=0)


def btle_capture(n_samples: int, channel: int = 37, seed: int = 1, mean_gap: float = 20000.0,
                 sigma: float = 0.05, cfo_max_hz: float = 50e3, fs: float = 4e6,
                 amplitude: float = 1.0, n_packets: Optional[int] = None,
                 tail_guard: int = 2048, noise: bool = True,
                 max_advdata: int = 31) -> Tuple[np.ndarray, List[TruthPacket]]:
    """Single-channel capture at 4 samples/symbol: AWGN everywhere + GFSK advertising packets
    separated by exponential gaps (SURVEY §8d cfg #2). Returns (complex64[n_samples], truth)."""
    rng = np.random.default_rng(seed)
    x = np.zeros(n_samples, dtype=np.complex64)
    truth: List[TruthPacket] = []
    pos = int(rng.exponential(mean_gap)) + 256
    pad = 4
    while True:
        if n_packets is not None and len(truth) >= n_packets:
            break
        pdu = btle_random_pdu(rng, max_advdata)
        bits = btle_air_bits(pdu, channel)
        wave = gfsk_modulate(bits, pad_symbols=pad)
        if pos + wave.size + tail_guard > n_samples:
            break
        cfo = rng.uniform(-cfo_max_hz, cfo_max_hz)
        ph0 = rng.uniform(0, 2 * math.pi)
        n = np.arange(wave.size)
        rot = np.exp(1j * (2 * math.pi * cfo / fs * n + ph0)).astype(np.complex64)
        x[pos:pos + wave.size] += (amplitude * wave * rot).astype(np.complex64)
        aa_start = pos + (pad + 8) * BTLE_SPS
        truth.append(TruthPacket(0, channel, aa_start, pdu, {"cfo": cfo}))
        pos += wave.size + int(rng.exponential(mean_gap))
    if noise and sigma > 0:
        x += (sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
              ).astype(np.complex64)
    return x, truth


def btle_capture_of(pdus: Sequence[bytes], spacing: int = 20000, channel: int = 37, seed: int = 1, sigma: float = 0.02,
                    first: int = 2048, fs: float = 4e6, cfo_max_hz: float = 20e3) -> Tuple[np.ndarray, List[TruthPacket]]:
    """A capture that carries exactly the given advertising PDUs, one every ``spacing`` samples from sample ``first``
    (small random CFO and phase per packet, AWGN everywhere): loopback input for tests that care about WHAT is sent."""
    rng = np.random.default_rng(seed)
    n_samples = first + spacing * len(pdus) + 4096
    x = np.zeros(n_samples, dtype=np.complex64)
    truth: List[TruthPacket] = []
    pad = 4
    for k, pdu in enumerate(pdus):
        wave = gfsk_modulate(btle_air_bits(pdu, channel), pad_symbols=pad)
        assert wave.size + 64 < spacing
        pos = first + k * spacing
        cfo = rng.uniform(-cfo_max_hz, cfo_max_hz)
        rot = np.exp(1j * (2 * math.pi * cfo / fs * np.arange(wave.size) + rng.uniform(0, 2 * math.pi))).astype(np.complex64)
        x[pos:pos + wave.size] += (wave * rot).astype(np.complex64)
        truth.append(TruthPacket(0, channel, pos + (pad + 8) * BTLE_SPS, pdu, {"cfo": cfo}))
    x += (sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))).astype(np.complex64)
    return x, truth


def to_interleaved(x: np.ndarray) -> np.ndarray:
    """complex64[n] -> float32[2n] view (re, im interleaved), the on-disk/ABI layout."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    return x.view(np.float32)


# ------------------------------------------------------------------------------------------------
# IEEE 802.15.4 O-QPSK (2.4 GHz PHY)
# ------------------------------------------------------------------------------------------------
ZB_BASE_CHIPS = "11011001110000110101001000101110"   # symbol 0, c0..c31 (IEEE 802.15.4-2003 Table 24)


def zb_chip_table() -> np.ndarray:
    """16 x 32 chip values (0/1). Symbols 1..7 are symbol 0 cyclically shifted right by 4 chips
    each; symbols 8..15 repeat 0..7 with the odd-indexed (Q) chips inverted."""
    base = np.array([int(c) for c in ZB_BASE_CHIPS], dtype=np.uint8)
    tab = np.zeros((16, 32), dtype=np.uint8)
    for k in range(8):
        tab[k] = np.roll(base, 4 * k)
        tab[k + 8] = tab[k]
        tab[k + 8][1::2] ^= 1
    return tab


def zb_chip_words() -> np.ndarray:
    """FM-domain (MSK) chip words: d_k = c_{k-1} xor c_k xor (k & 1), first chip in the MSB.
    Equal to gr-ieee802-15-4's CHIP_MAPPING under the mask 0x7FFFFFFE (c_{-1} taken as 0)."""
    tab = zb_chip_table()
    words = np.zeros(16, dtype=np.uint32)
    for s in range(16):
        prev = 0
        w = 0
        for k in range(32):
            d = prev ^ int(tab[s][k]) ^ (k & 1)
            w = (w << 1) | d
            prev = int(tab[s][k])
        words[s] = w
    return words


def crc16_154(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x8408 if c & 1 else c >> 1
    return c


def zb_frame(psdu_wo_fcs: bytes) -> bytes:
    """PSDU = payload + FCS (CRC-16 ITU-T, little-endian)."""
    c = crc16_154(psdu_wo_fcs)
    return bytes(psdu_wo_fcs) + bytes([c & 0xFF, c >> 8])


def oqpsk_modulate(psdu: bytes, tail_symbols: int = 2) -> np.ndarray:
    """SHR (4 x 0x00, SFD 0xA7) + PHR + PSDU -> complex64 at 4 samples per chip pair, following the
    reference transmitter (Zigbee_tx/top_block.py:59-71): nibble (low first) -> 16 complex chips
    -> repeat 4 -> x [0, sin(pi/4), 1, sin(3pi/4)] -> Q delayed by 2 samples."""
    assert len(psdu) <= 127
    ppdu = bytes([0, 0, 0, 0, 0xA7, len(psdu)]) + bytes(psdu)
    tab = zb_chip_table()
    chips = []
    for b in ppdu:
        chips.append(tab[b & 0xF])
        chips.append(tab[b >> 4])
    chips = np.concatenate(chips).astype(np.float64) * 2.0 - 1.0
    i_ch = chips[0::2]
    q_ch = chips[1::2]
    shape = np.array([0.0, math.sin(math.pi / 4), 1.0, math.sin(3 * math.pi / 4)])
    n = i_ch.size * 4 + 2 + 4 * 16 * tail_symbols
    i_s = np.zeros(n)
    q_s = np.zeros(n)
    i_s[:i_ch.size * 4] = np.repeat(i_ch, 4) * np.tile(shape, i_ch.size)
    q_s[2:2 + q_ch.size * 4] = np.repeat(q_ch, 4) * np.tile(shape, q_ch.size)
    return (i_s + 1j * q_s).astype(np.complex64)


ZB_SLOT = 17408         # samples of one timeslot at 4 Msps: the longest PPDU (133 B = 17 024 samples), tail, guard


def zigbee_capture(n_samples: int, channel: int = 11, seed: int = 4, mean_gap: float = 20000.0,
                   sigma: float = 0.05, cfo_max_hz: float = 50e3, fs: float = 4e6,
                   amplitude: float = 1.0, n_packets: Optional[int] = None, min_len: int = 5,
                   max_len: int = 127, noise: bool = True, tail_guard: int = 4096,
                   slot_phase: Optional[int] = None, slot_jitter: int = 0) -> Tuple[np.ndarray, List[TruthPacket]]:
    """Single 802.15.4 channel at 4 Msps (2 samples/chip): AWGN + frames with valid FCS separated
    by exponential gaps (SURVEY §8d cfg #4, per channel).

    ``slot_phase`` (0 or 1): slotted traffic as in a TSCH schedule -- time is cut into timeslots of
    ZB_SLOT samples, this channel transmits only in the slots of its phase (even / odd), one frame
    per used slot, starting at the slot boundary + 64 samples (+ a random 0 .. ``slot_jitter`` - 1 samples: without it every
    transmitter of a slotted capture shares ONE chip clock phase, which no set of real radios does -- a receiver's
    timing loop then stays locked from one frame, even a neighbour's leakage, to the next); slots are used with the
    probability that keeps the mean frame rate of the unslotted model."""
    rng = np.random.default_rng(seed)
    x = np.zeros(n_samples, dtype=np.complex64)
    truth: List[TruthPacket] = []
    if slot_phase is not None:
        mean_len = 128.0 * (6 + 0.5 * (min_len + max_len))      # samples of a mean PPDU (2 symbols of 64 per byte)
        p_use = min(1.0, 2.0 * ZB_SLOT / (mean_gap + mean_len))
        k = 0
        while True:
            start = (2 * k + int(slot_phase)) * ZB_SLOT + 64
            k += 1
            if start + ZB_SLOT + tail_guard > n_samples or (n_packets is not None and len(truth) >= n_packets):
                break
            if rng.random() >= p_use:
                continue
            if slot_jitter:
                start += int(rng.integers(0, slot_jitter))
            ln = int(rng.integers(min_len, max_len + 1))
            psdu = zb_frame(bytes(rng.integers(0, 256, ln - 2, dtype=np.uint8)))
            wave = oqpsk_modulate(psdu)
            cfo = rng.uniform(-cfo_max_hz, cfo_max_hz)
            ph0 = rng.uniform(0, 2 * math.pi)
            nn = np.arange(wave.size)
            rot = np.exp(1j * (2 * math.pi * cfo / fs * nn + ph0)).astype(np.complex64)
            x[start:start + wave.size] += (amplitude * wave * rot).astype(np.complex64)
            truth.append(TruthPacket(1, channel, start, psdu, {"cfo": cfo}))
        if noise and sigma > 0:
            x += (sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
                  ).astype(np.complex64)
        return x, truth
    pos = int(rng.exponential(mean_gap)) + 512
    while True:
        if n_packets is not None and len(truth) >= n_packets:
            break
        ln = int(rng.integers(min_len, max_len + 1))
        psdu = zb_frame(bytes(rng.integers(0, 256, ln - 2, dtype=np.uint8)))
        wave = oqpsk_modulate(psdu)
        if pos + wave.size + tail_guard > n_samples:
            break
        cfo = rng.uniform(-cfo_max_hz, cfo_max_hz)
        ph0 = rng.uniform(0, 2 * math.pi)
        nn = np.arange(wave.size)
        rot = np.exp(1j * (2 * math.pi * cfo / fs * nn + ph0)).astype(np.complex64)
        x[pos:pos + wave.size] += (amplitude * wave * rot).astype(np.complex64)
        truth.append(TruthPacket(1, channel, pos, psdu, {"cfo": cfo}))
        pos += wave.size + int(rng.exponential(mean_gap))
    if noise and sigma > 0:
        x += (sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
              ).astype(np.complex64)
    return x, truth


# ------------------------------------------------------------------------------------------------
# Wideband compositor
# ------------------------------------------------------------------------------------------------
def _upsample_to_wideband(x: np.ndarray, up: int) -> np.ndarray:
    """Integer-rate upsampling of a 4 Msps channel to the wideband rate (zero-stuff + windowed-sinc
    low-pass, scipy.signal.resample_poly)."""
    from scipy.signal import resample_poly
    return resample_poly(x.astype(np.complex128), up, 1).astype(np.complex64)


def btle_bin_channel(b: int) -> int:
    """BLE channel index carried by bin b of the M = 40 channelizer centred at 2442 MHz."""
    k = (b + 20) % 40
    if k == 0:
        return 37
    if k == 12:
        return 38
    if k == 39:
        return 39
    return k - 1 if k <= 11 else k - 2


def zigbee_bin_channel(b: int) -> int:
    return 11 + (b + 8) % 16


def wideband_capture(proto: int, n_samples: int, seed: int = 3, bins: Optional[Sequence[int]] = None,
                     mean_gap: float = 20000.0, sigma: float = 0.05, cfo_max_hz: float = 50e3,
                     max_len: int = 127, slotted: Optional[bool] = None, slot_jitter: int = 0) -> Tuple[np.ndarray, List[TruthPacket]]:
    """Wideband synthetic capture (SURVEY §8d cfg #3 / #4): every listed channelizer bin carries an
    independent narrowband 4 Msps traffic stream, upsampled by M/2... i.e. to fs = M * 2 MHz, shifted
    to its bin centre (bin b -> b fs / M, wrapping) and summed; AWGN added at the wideband rate.
    proto 0: BTLE, M = 40, fs = 80 Msps.  proto 1: 802.15.4, M = 16, fs = 32 Msps."""
    M = 40 if proto == 0 else 16
    up = M // 2
    bins = list(range(M)) if bins is None else list(bins)
    n_ch = n_samples // up
    x = np.zeros(n_samples, dtype=np.complex64)
    truth: List[TruthPacket] = []
    t = np.arange(n_ch * up)
    for b in bins:
        if proto == 0:
            ch = btle_bin_channel(b)
            nb, tr = btle_capture(n_ch, channel=ch, seed=seed * 1000 + b, mean_gap=mean_gap,
                                  noise=False, cfo_max_hz=cfo_max_hz)
        else:
            ch = zigbee_bin_channel(b)
            # cfg #4's synthetic 2 MHz raster puts a 2 Mchip/s O-QPSK signal (main lobe +-1.5 MHz) on every
            # bin: neighbours overlap spectrally, so their traffic is scheduled TSCH-style -- even and odd
            # bins transmit in alternating timeslots (DESIGN.md deviation 7); `slotted=False` is the
            # unscheduled model, where colliding neighbours lose frames on any receiver
            use_slots = (len(bins) > 8) if slotted is None else slotted
            nb, tr = zigbee_capture(n_ch, channel=ch, seed=seed * 1000 + b, mean_gap=mean_gap,
                                    noise=False, cfo_max_hz=cfo_max_hz, max_len=max_len,
                                    slot_phase=(b & 1) if use_slots else None, slot_jitter=slot_jitter)
        wb = _upsample_to_wideband(nb, up)
        rot = np.exp(2j * np.pi * ((b * t) % M) / M).astype(np.complex64)
        x[:wb.size] += wb * rot
        truth.extend(tr)
    if sigma > 0:
        rng = np.random.default_rng(seed)
        x += (sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
              ).astype(np.complex64)
    return x, truth


def quantize(x: np.ndarray, sample_format: int, full_scale: float = 0.0) -> np.ndarray:
    """Complex capture -> interleaved int8 (sample_format 1, HackRF) or int16 (2, USRP sc16) as an
    SDR's ADC path would deliver it: scaled so that ``full_scale`` (default: 1.25 x the largest
    component) maps to the integer range, rounded to nearest, clipped."""
    a = to_interleaved(x)
    bits = {1: 7, 2: 15}[sample_format]
    fs = full_scale or 1.25 * float(np.max(np.abs(a))) or 1.0
    q = np.rint(a * ((1 << bits) / fs))
    lim = (1 << bits) - 1
    return np.clip(q, -lim - 1, lim).astype(np.int8 if sample_format == 1 else np.int16)
