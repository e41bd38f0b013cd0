"""Dev tool (CPU; the oracle = what the GPU computes bit for bit): which 802.15.4 lane shape cfg #5's segments should run
with -- frames lost / extra against ONE sequential lane per channel, for 2^24-input-sample segments of bench.py's cfg #4 / #5
traffic (all 16 bins busy, slotted, AWGN sigma 0.05 on top), core 2048 / 4096 / 8192, warm-up 512.
Writes the table of profiles/r4_lane_residual.md (the ms/step column comes from the GPU runs quoted there).

    python tools/lane_residual_r4.py [segments, default 4]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from snout_amd import synth
from oracle import oracle_py

SEG = 1 << 24


def key(a):
    return [(int(c), bytes(b[:l]), int(s), int(k)) for c, s, l, b, k in zip(a["channel"], a["sample_index"], a["len"], a["bytes"], a["crc_ok"])]


def missing(P, Q):
    d = collections.defaultdict(list)
    for c, b, s, k in Q:
        d[(c, b)].append(s)
    return [i for i, (c, b, s, k) in enumerate(P) if not any(abs(s - u) <= 8 for u in d.get((c, b), []))]


if __name__ == "__main__":
    nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    oracle_py.set_threads(os.cpu_count())
    tile, truth = synth.wideband_capture(1, SEG // 8, seed=4, sigma=0.0)         # bench.py make_tile("cfg4", n_tile = seg / 8)
    tot = collections.Counter()
    shapes = [(2048, 512), (4096, 512), (8192, 512), (4096, 1024)]
    for sg in range(nseg):
        rng = np.random.default_rng(100 + sg)
        x = np.tile(tile, 8)
        x = (x + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        one = key(oracle_py.wideband_segment(x, proto=1, core=1 << 22, warmup=512))
        tot["one"] += len(one)
        tot["one_ok"] += sum(k for *_, k in one)
        tot["sent"] += 8 * len(truth)
        for core, warm in shapes:
            rec = oracle_py.wideband_segment(x, proto=1, core=core, warmup=warm)
            lan = key(rec)
            tot[(core, warm, "n")] += len(lan)
            tot[(core, warm, "ok")] += sum(k for *_, k in lan)
            tot[(core, warm, "lost")] += len(missing(one, lan))
            tot[(core, warm, "extra")] += len(missing(lan, one))
            tot[(core, warm, "flag")] += int(((rec["flags"] & 4) != 0).sum())
        print("segment", sg, dict((str(k), v) for k, v in tot.items()), flush=True)
    print()
    print("| lane shape (core / warm-up) | lane work | frames | FCS ok | lost vs one lane | extra | flagged SEAM_DISAGREED |")
    print("|---|---|---|---|---|---|---|")
    print("| one lane per channel (the sequential receiver) | 1.00 | %d | %d | - | - | 0 |" % (tot["one"], tot["one_ok"]))
    for core, warm in shapes:
        print("| %d / %d | %.3f | %d | %d | %d (%.2f %%) | %d | %d |" % (core, warm, (core + warm) / core, tot[(core, warm, "n")], tot[(core, warm, "ok")],
              tot[(core, warm, "lost")], 100.0 * tot[(core, warm, "lost")] / tot["one"], tot[(core, warm, "extra")], tot[(core, warm, "flag")]))
    print("\n%d segments of 2^24 input samples, %d frames transmitted" % (nseg, tot["sent"]))
