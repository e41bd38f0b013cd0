"""Randomised GPU == oracle parity sweep (developer tool): random sizes, lane shapes, densities and
impairments for the four receive paths.  Prints the failing case and stops at the first mismatch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
from oracle import oracle_py as oracle

FIELDS = ("sample_index", "proto", "channel", "len", "crc_ok", "lqi", "pdu_type", "flags", "aux")


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in FIELDS) and np.array_equal(a["bytes"], b["bytes"])


def main(budget_s):
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    t_end = time.time() + budget_s
    n_cases = {"btle": 0, "zigbee": 0, "btle40": 0, "zigbee16": 0}
    n_repaired = [0]
    while time.time() < t_end:
        kind = rng.choice(list(n_cases))
        seed = int(rng.integers(1 << 30))
        first = int(rng.integers(0, 1 << 40))
        fmt = int(rng.choice([0, 0, 1, 2]))       # cf32 / sc8 / sc16 input

        def prep(x):
            """-> (what the receiver gets, what the oracle gets)"""
            if fmt == 0:
                return x, x
            q = synth.quantize(x, fmt, full_scale=float(rng.choice([0.0, 0.5, 4.0])))   # incl. clipping / coarse
            return q, oracle.from_int(q)
        if kind == "btle":
            n = int(rng.integers(5, 1 << 19))
            ch = int(rng.integers(0, 40))
            x, _ = synth.btle_capture(n, channel=ch, seed=seed, mean_gap=float(rng.choice([300.0, 3000.0, 30000.0])))
            xin, x = prep(x)
            with SnoutRx(proto=0, channel=ch, sample_format=fmt) as rx:
                got = rx.process(xin, first_sample_index=first)
            want, _ = oracle.btle_segment(x, channel=ch, first_sample_index=first)
            desc = f"btle n={n} ch={ch} seed={seed} fmt={fmt}"
        elif kind == "zigbee":
            n = int(rng.integers(9, 1 << 19))
            core = int(rng.choice([1024, 2048, 4096, 6144, 8192, 16384]))
            warm = int(rng.choice([w for w in (256, 512, 1024, 2048, 3072, 4096) if w < core]))
            # (heavy noise now and then: frames that the lanes' sinks give up behind a seam, i.e. frame repairs)
            x, _ = synth.zigbee_capture(n, seed=seed, mean_gap=float(rng.choice([400.0, 4000.0, 20000.0])),
                                        cfo_max_hz=float(rng.choice([0.0, 40e3, 100e3])), sigma=float(rng.choice([0.02, 0.05, 0.3, 0.5])))
            if rng.random() < 0.2 and fmt == 0:
                x[int(rng.integers(0, n))] = np.nan
            xin, x = prep(x)
            with SnoutRx(proto=1, channel=11, zb_core=core, zb_warmup=warm, sample_format=fmt) as rx:
                got = rx.process(xin, first_sample_index=first)
            want = oracle.zigbee_segment(x, channel=11, core=core, warmup=warm, first_sample_index=first)
            desc = f"zigbee n={n} core={core} warm={warm} seed={seed} fmt={fmt}"
        elif kind == "btle40":
            n = int(rng.integers(640, 40 * 60000))
            x, _ = synth.wideband_capture(0, n, seed=seed, bins=sorted(rng.choice(40, 6, replace=False).tolist()),
                                          mean_gap=float(rng.choice([2000.0, 8000.0])))
            x = x[:n]
            if rng.random() < 0.2 and fmt == 0:             # non-finite samples: 16 taps' reach, nothing more
                x[int(rng.integers(0, n))] = [np.nan, np.inf, -np.inf][int(rng.integers(3))]
            xin, x = prep(x)
            with SnoutRx(proto=0, n_channels=40, sample_format=fmt) as rx:
                got = rx.process(xin, first_sample_index=first)
            want = oracle.wideband_segment(x, 0, first_sample_index=first)
            desc = f"btle40 n={n} seed={seed} fmt={fmt}"
        else:
            n = int(rng.integers(256, 16 * 60000))
            core = int(rng.choice([0, 1024, 2048, 4096]))       # 0: the default shape
            # four bins, or every bin busy with noise on top (cfg #4's traffic: the frame repair at work)
            dense = rng.random() < 0.4
            x, _ = synth.wideband_capture(1, n, seed=seed, bins=None if dense else sorted(rng.choice(16, 4, replace=False).tolist()),
                                          mean_gap=float(rng.choice([3000.0, 12000.0])), max_len=int(rng.choice([40, 127])),
                                          sigma=float(rng.choice([0.0, 0.05, 0.1])) if dense else 0.05)
            x = x[:n]
            if rng.random() < 0.2 and fmt == 0:
                x[int(rng.integers(0, n))] = [np.nan, np.inf, -np.inf][int(rng.integers(3))]
            xin, x = prep(x)
            with SnoutRx(proto=1, n_channels=16, zb_core=core, sample_format=fmt) as rx:
                got = rx.process(xin, first_sample_index=first)
            want = oracle.wideband_segment(x, 1, first_sample_index=first, core=core)
            desc = f"zigbee16 n={n} core={core} seed={seed} fmt={fmt} dense={dense}"
        n_cases[kind] += 1
        if kind.startswith("zigbee"):
            n_repaired[0] += int(((want["flags"] & 8) != 0).sum())
        if not same(got, want):
            print("MISMATCH:", desc, "first_index", first, len(got), len(want), flush=True)
            return 1
    print("fuzz ok:", n_cases, "repaired frames in the 802.15.4 cases:", n_repaired[0], flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0))
