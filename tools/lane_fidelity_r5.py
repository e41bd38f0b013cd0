"""Dev tool (CPU; the oracle = what the GPU computes record for record): the 802.15.4 lanes, with and without the frame repair,
against ONE sequential lane per channel on cfg #4's dense traffic (all 16 bins busy), on the same traffic with the frames' starts
jittered, or on a sparse capture (8 even bins).  -> profiles/r5_lane_fidelity.md

    SHAPES="6144,1024;8192,1024" python tools/lane_fidelity_r5.py dense|jitter|sparse [segments, default 4]"""
import collections, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from snout_amd import synth
from oracle import oracle_py

SEG = 1 << 24


def key(a):
    return [(int(c), bytes(b[:l]), int(s)) for c, s, l, b in zip(a["channel"], a["sample_index"], a["len"], a["bytes"])]


def missing(P, Q):
    d = collections.defaultdict(list)
    for c, b, s in Q:
        d[(c, b)].append(s)
    return sum(1 for c, b, s in P if not any(abs(s - u) <= 8 for u in d.get((c, b), [])))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "dense"
    nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("SHAPES", "4096,512;6144,1024;8192,1024").split(";")]
    oracle_py.set_threads(os.cpu_count())
    lib = oracle_py.lib()
    if which == "dense":
        tile, truth = synth.wideband_capture(1, SEG // 8, seed=4, sigma=0.0)
    elif which == "jitter":
        tile, truth = synth.wideband_capture(1, SEG // 8, seed=4, sigma=0.0, slot_jitter=32)
    else:
        tile, truth = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
    tot = collections.Counter()
    for sg in range(nseg):
        rng = np.random.default_rng(100 + sg)
        x = np.tile(tile, 8)
        x = (x + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        one = key(oracle_py.wideband_segment(x, proto=1, core=1 << 22, warmup=512))
        tot["one"] += len(one)
        tot["sent"] += 8 * len(truth)
        for core, warm in shapes:
            for rep in (0, 1):
                lib.oracle_zb_set_repair(rep)
                rec = oracle_py.wideband_segment(x, proto=1, core=core, warmup=warm)
                lan = key(rec)
                tot[(core, warm, rep, "lost")] += missing(one, lan)
                tot[(core, warm, rep, "extra")] += missing(lan, one)
                tot[(core, warm, rep, "rep")] += int(((rec["flags"] & 8) != 0).sum())
        lib.oracle_zb_set_repair(1)
    print(f"{which} capture, {nseg} segments of 2^24 input samples: {tot['sent']} frames sent, {tot['one']} decoded by one sequential lane per channel")
    print("| core / warm-up | | lost vs one lane | extra | lost + extra | repaired |\n|---|---|---|---|---|---|")
    for core, warm in shapes:
        for rep in (0, 1):
            k = (core, warm, rep)
            lo, ex = tot[k + ("lost",)], tot[k + ("extra",)]
            print(f"| {core} / {warm} | {'lanes + frame repair' if rep else 'lanes alone (round 4)'} | {lo} ({100.0 * lo / tot['one']:.2f} %) | "
                  f"{ex} ({100.0 * ex / tot['one']:.2f} %) | {100.0 * (lo + ex) / tot['one']:.2f} % | {tot[k + ('rep',)]} |")
