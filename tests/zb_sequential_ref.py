"""A second, independently written statement of the 802.15.4 receive chain (TEST INFRASTRUCTURE).

Written in numpy / plain Python straight from the block descriptions in SURVEY.md Appendix A.2.1-A.2.4
(GNU Radio 3.7 `quadrature_demod_cf` -> `single_pole_iir_filter_ff` + `sub_ff` ->
`clock_recovery_mm_ff` -> gr-ieee802-15-4 `packet_sink`), with the parameters of the reference
flowgraph (snout/modulations/Zigbee/hackrf/Zigbee_rx/top_block.py:52,67,69,70,73): one continuous
stream, one loop, one sink -- no lanes, no stitching, no carry-in.  It exists so that the C oracle
(`oracle/oracle_zigbee.c`, run with one lane) is not the only statement of the M&M loop and the sink
state machine: `tests/test_zigbee_sequential.py` compares the two record for record.

Only data is shared with the oracle: the MMSE interpolator bank (`mmse_taps.inc`, regenerated offline,
DESIGN.md deviation 4) and the 16 chip words (pinned to the reference's TX table by
tests/test_oracle_zigbee.py).  Float conventions: every f32 product and sum is rounded separately, as
GNU Radio's scalar C++ does; the 8-tap interpolation is an fma chain (the oracle's stated choice),
evaluated here in 80-bit and rounded once.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32
ALPHA = 0.00016                                     # single_pole_iir_filter_ff(0.00016, 1)  top_block.py:52
OMEGA, GAIN_OMEGA, MU0, GAIN_MU, OMEGA_REL = 2.0, 0.000225, 0.5, 0.03, 0.0002     # top_block.py:69
MASK = 0x7FFFFFFE


def quad_demod(iq: np.ndarray) -> np.ndarray:
    """A.2.1: y[n] = fast_atan2f(Im(x[n] conj x[n-1]), Re(..)), x[-1] = 0; table of atan(i/255)."""
    x = np.ascontiguousarray(iq, dtype=np.complex64)
    re, im = x.real.astype(F), x.imag.astype(F)
    pre = np.concatenate([[F(0)], re[:-1]])
    pim = np.concatenate([[F(0)], im[:-1]])
    with np.errstate(all="ignore"):
        pr = (re * pre).astype(F) + (im * pim).astype(F)          # Re
        pi_ = (im * pre).astype(F) - (re * pim).astype(F)         # Im
        tab = np.arctan(np.arange(257, dtype=np.float64) / 255.0).astype(F)
        ya, xa = np.abs(pi_), np.abs(pr)
        small_is_y = ya < xa
        z = np.where(small_is_y, ya / xa, xa / ya).astype(F)
        a = (z * F(255.0)).astype(F)
        k = np.nan_to_num(a, nan=0.0, posinf=0.0, neginf=0.0).astype(np.int64) & 0xFF
        frac = (a - k.astype(F)).astype(F)
        interp = (tab[k] + ((tab[k + 1] - tab[k]).astype(F) * frac).astype(F)).astype(F)
        base = np.where(z < F(0.003921569), z, interp).astype(F)
        PI, H = F(math.pi), F(math.pi / 2)
        ang_x = np.where(pr >= 0, np.where(pi_ >= 0, base, -base),
                         np.where(pi_ >= 0, PI - base, base - PI))
        ang_y = np.where(pi_ >= 0, np.where(pr >= 0, H - base, H + base),
                         np.where(pr >= 0, -H + base, -H - base))
        ang = np.where(xa > ya, ang_x, ang_y).astype(F)
        ang = np.where((ya > 0) | (xa > 0), ang, F(0)).astype(F)
        ang = np.where(np.abs(ang) <= F(4.0), ang, F(0)).astype(F)   # non-finite input -> 0 (DESIGN deviation 5)
    return ang


def dc_removed(d: np.ndarray) -> np.ndarray:
    """A.2.2: z[n] = d[n] - float(lp[n]), lp[n] = a d[n] + (1-a) lp[n-1] in double, lp[-1] = 0."""
    out = np.empty(d.size, dtype=F)
    lp, a, b = 0.0, ALPHA, 1.0 - ALPHA
    for i, v in enumerate(d.tolist()):
        lp = a * v + b * lp
        out[i] = F(v) - F(lp)
    return out


def clock_recovery(z: np.ndarray, taps: np.ndarray):
    """A.2.3 with (omega, gain_omega, mu, gain_mu, limit) = (2, 0.000225, 0.5, 0.03, 0.0002).
    Returns (chip values f32, window start of every chip)."""
    n = z.size
    tl = taps.astype(np.longdouble)
    zl = z.astype(np.longdouble)
    mu, omega, last = F(MU0), F(OMEGA), F(0.0)
    mid, lim = F(OMEGA), F(F(OMEGA) * F(OMEGA_REL))
    g_om, g_mu = F(GAIN_OMEGA), F(GAIN_MU)
    chips, where = [], []
    ii = 0
    while ii + 8 <= n:
        imu = int(np.rint(F(mu * F(128.0))))
        acc = np.longdouble(0)
        row = tl[imu]
        for k in range(8):                          # sum_k taps[imu][k] * in[ii + 7 - k], one rounding per step
            acc = np.longdouble(F(row[k] * zl[ii + 7 - k] + acc))
        o = F(acc)
        chips.append(o)
        where.append(ii)
        s_last = F(-1.0) if last < 0 else F(1.0)
        s_o = F(-1.0) if o < 0 else F(1.0)
        mm = F(F(s_last * o) - F(s_o * last))
        last = o
        omega = F(omega + F(g_om * mm))
        x = F(omega - mid)
        clipped = F(F(0.5) * F(abs(F(x + lim)) - abs(F(x - lim))))      # branchless_clip(x, lim)
        omega = F(mid + clipped)
        mu = F(F(mu + omega) + F(g_mu * mm))
        fl = math.floor(float(mu))
        ii += fl if fl >= 1 else 1
        mu = F(mu - F(fl))
    return np.array(chips, dtype=F), np.array(where, dtype=np.int64)


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x8408 if c & 1 else c >> 1
    return c


def packet_sink(chips: np.ndarray, where: np.ndarray, words, threshold: int = 10):
    """A.2.4.  Returns (chip index of the chip that completed the SFD, PSDU bytes, lqi) per frame."""
    words = [int(w) & MASK for w in words]

    def dist(reg, s):
        return bin((reg & MASK) ^ words[s]).count("1")

    def best(reg):
        d = [dist(reg, s) for s in range(16)]
        m = min(d)
        return d.index(m), m

    q, n = 0, chips.size
    reg = 0
    frames = []
    while q < n:
        # ---- search for the first zero symbol, chip by chip
        reg = ((reg << 1) | (1 if chips[q] > 0 else 0)) & 0xFFFFFFFF
        q += 1
        if dist(reg, 0) >= threshold:
            continue
        trigger = q - 1
        # ---- preamble zeros, then 0x7, then 0xA, one symbol (32 chips) at a time
        state = "zeros"
        ok = False
        while q + 32 <= n:
            for _ in range(32):
                reg = ((reg << 1) | (1 if chips[q] > 0 else 0)) & 0xFFFFFFFF
                q += 1
            if state == "zeros":
                if dist(reg, 0) <= threshold:
                    continue
                if dist(reg, 7) <= threshold:
                    state = "sfd"
                    continue
                break
            ok = dist(reg, 0xA) <= threshold
            sync = q - 1                              # the chip that completed the SFD
            break
        else:
            # fewer than 32 chips left: consume them (the stream ends inside the frame)
            while q < n:
                reg = ((reg << 1) | (1 if chips[q] > 0 else 0)) & 0xFFFFFFFF
                q += 1
            break
        if not ok:
            reg = 0                                   # back to SYNC_SEARCH with a cleared register
            continue
        # ---- PHR (2 symbols, low nibble first) and the PSDU; LQI over the first 8 decoded symbols
        lqi_sum, lqi_n = 0, 0
        nibbles = []
        need = 2
        length = None
        dead = False
        while len(nibbles) < need:
            if q + 32 > n:
                dead = True
                q = n
                break
            for _ in range(32):
                reg = ((reg << 1) | (1 if chips[q] > 0 else 0)) & 0xFFFFFFFF
                q += 1
            s, m = best(reg)
            if m >= threshold:
                dead = True
                break
            if lqi_n < 8:
                lqi_sum += 32 - m
                lqi_n += 1
            nibbles.append(s)
            if length is None and len(nibbles) == 2:
                length = nibbles[0] | (nibbles[1] << 4)
                if length > 127:
                    dead = True
                    break
                nibbles = []
                need = 2 * max(length, 1)             # the sink publishes after the first byte even if PHR = 0
        reg_keep = reg
        reg = 0
        if dead or length is None:
            continue
        data = bytes(nibbles[2 * i] | (nibbles[2 * i + 1] << 4) for i in range(len(nibbles) // 2))
        lqi = min(255, (lqi_sum // 8) << 3)
        frames.append((sync, data, lqi))
        del reg_keep
    return frames


def receive(iq: np.ndarray, taps: np.ndarray, words, channel: int = 11, threshold: int = 10,
            first_index: int = 0):
    """The whole sequential chain -> list of dicts with the fields of a `snout_pkt` record."""
    d = quad_demod(iq)
    z = dc_removed(d)
    chips, where = clock_recovery(z, taps)
    out = []
    for sync, data, lqi in packet_sink(chips, where, words, threshold):
        ok = 0
        if len(data) >= 3:
            c = crc16(data[:-2])
            ok = int((c & 0xFF) == data[-2] and (c >> 8) == data[-1])
        # the record's sample_index: window start of the chip 319 chips before the SFD-completing one,
        # i.e. the first chip of a regular preamble + SFD (oracle_zigbee.c "sample_index")
        out.append({"sample_index": first_index + int(where[max(sync - 319, 0)]), "channel": channel, "len": len(data),
                    "lqi": lqi, "crc_ok": ok, "bytes": data})
    return out, d, z, chips
