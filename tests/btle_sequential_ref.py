"""A second, independently written statement of the BTLE receive chain (TEST INFRASTRUCTURE).

Plain numpy / Python from SURVEY.md Appendix A.1 (`btle_rx`: `search_unique_bits` -> `demod_byte` ->
`scramble_byte` -> header -> `crc_check`), called by the reference as `btle_rx -c CH -g 6 -a 8e89bed6
-k 555555` (snout/util/btle.py:63-68).  It shares no code with `oracle/oracle_btle.c`: the hard bits are
one vectorised comparison, the access-address matches of all four sampling phases come from a sliding
window over the bit array, whitening and CRC-24 are `snout_amd.synth`'s bit-serial forms (straight from
the Bluetooth Core Spec), and only the resume rule is a loop.  `tests/test_btle_sequential.py` compares it
with the C oracle record for record.
"""
from __future__ import annotations

import numpy as np

from snout_amd import synth


def hard_bits(iq: np.ndarray) -> np.ndarray:
    """bit[n] = (I[n] Q[n+4]) > (I[n+4] Q[n]), each product rounded to f32 (A.1: sign of I0 Q1 - I1 Q0)."""
    x = np.ascontiguousarray(iq, dtype=np.complex64)
    i, q = x.real.astype(np.float32), x.imag.astype(np.float32)
    with np.errstate(all="ignore"):
        return ((i[:-4] * q[4:]).astype(np.float32) > (i[4:] * q[:-4]).astype(np.float32)).astype(np.uint8)


def aa_matches(bits: np.ndarray, aa: int) -> np.ndarray:
    """Samples n (ascending) at which the last 32 bits of n's sampling phase, oldest first, spell the
    access address LSB first: bit k of aa == bits[n - 4 (31 - k)]."""
    nb = bits.size
    if nb < 125:
        return np.zeros(0, dtype=np.int64)
    ok = np.ones(nb - 124, dtype=bool)                 # candidate n = 124 + t
    for k in range(32):
        ok &= bits[4 * k:4 * k + nb - 124] == ((aa >> k) & 1)
    return np.nonzero(ok)[0] + 124


def _byte(bits, at):
    return int(sum(int(bits[at + 4 * k]) << k for k in range(8)))


def receive(iq: np.ndarray, channel: int = 37, aa: int = 0x8E89BED6, crc_init: int = 0x555555,
            first_index: int = 0):
    bits = hard_bits(iq)
    nb = bits.size
    hits = aa_matches(bits, aa)
    wh = np.packbits(synth.btle_whiten_bits(channel, 42 * 8), bitorder="little")
    out = []
    resume = 0                       # the search after a packet starts with empty phase registers here
    pos = 0
    while True:
        pos = int(np.searchsorted(hits, resume + 124, side="left"))     # 32 bits of the phase since `resume`
        if pos >= hits.size:
            break
        n = int(hits[pos])
        hdr = n + 4
        if hdr + 60 >= nb:           # the header does not fit
            resume = n + 1
            continue
        h0 = _byte(bits, hdr) ^ int(wh[0])
        h1 = _byte(bits, hdr + 32) ^ int(wh[1])
        plen = h1 & 0x3F
        resume = hdr + 64
        if not 6 <= plen <= 37:
            continue
        total = 2 + plen + 3
        if hdr + 4 * (8 * total - 1) >= nb:      # the packet runs past the segment
            continue
        rec = bytes([h0, h1] + [_byte(bits, hdr + 32 * b) ^ int(wh[b]) for b in range(2, total)])
        crc_bits = synth.btle_crc24_bits(synth.bytes_to_bits_lsb(rec[:2 + plen]), crc_init)
        crc_rx = synth.bytes_to_bits_lsb(rec[2 + plen:]).tolist()
        out.append({"sample_index": first_index + n - 124, "channel": channel, "len": total,
                    "crc_ok": int(crc_bits == crc_rx), "pdu_type": h0 & 0x0F,
                    "flags": ((h0 >> 6) & 1) | (((h0 >> 7) & 1) << 1), "aux": n & 3, "bytes": rec})
        resume = hdr + 32 * total
    return out, bits
