"""Analysis tool (CPU, oracle only): the 802.15.4 lanes with their boundaries moved into the gaps between frames
(ORACLE_ZB_EXPERIMENT_GAPS in oracle/oracle_zigbee.c) against one sequential lane, on cfg #4's dense traffic (all 16 bins
busy) and on a sparse capture.  -> profiles/r4_lane_residual.md section 2.

    python tools/lane_gaps_r4.py [dense|sparse]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snout_amd import synth
from oracle import oracle_py

oracle_py.set_threads(os.cpu_count())
which = sys.argv[1] if len(sys.argv) > 1 else "dense"
if which == "dense":
    tile, _ = synth.wideband_capture(1, (1 << 24) // 8, seed=4, sigma=0.0, slot_jitter=4)
    reps, n_seg, sigma = 8, 2, 0.05
else:
    tile, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
    reps, n_seg, sigma = 16, 3, 0.05
key = lambda a: [(int(c), bytes(b[:l]), int(s)) for c, s, l, b in zip(a["channel"], a["sample_index"], a["len"], a["bytes"])]


def missing(A, B):
    d = collections.defaultdict(list)
    for c, b, s in B:
        d[(c, b)].append(s)
    return sum(1 for c, b, s in A if not any(abs(s - u) <= 8 for u in d.get((c, b), [])))


def run(x, core, warm, gaps, two=0):
    for k in ("ORACLE_ZB_EXPERIMENT_GAPS", "ORACLE_ZB_EXPERIMENT_TWOSTART"):
        os.environ.pop(k, None)
    if gaps:
        os.environ["ORACLE_ZB_EXPERIMENT_GAPS"] = str(gaps)
    if two:
        os.environ["ORACLE_ZB_EXPERIMENT_TWOSTART"] = str(two)
    r = oracle_py.wideband_segment(x, proto=1, core=core, warmup=warm)
    for k in ("ORACLE_ZB_EXPERIMENT_GAPS", "ORACLE_ZB_EXPERIMENT_TWOSTART"):
        os.environ.pop(k, None)
    return key(r[r["crc_ok"] == 1])


cases = [(2048, 512, 0, 0), (2048, 512, 0, 64), (2048, 512, 0, 128), (4096, 512, 0, 0), (4096, 512, 0, 32), (4096, 512, 0, 64), (4096, 512, 0, 128), (4096, 512, 0, 240),
         (4096, 256, 0, 64), (4096, 1024, 0, 128), (8192, 512, 0, 0), (8192, 512, 0, 128)]
if os.environ.get("GAPS"):
    cases = [(4096, 512, 0, 0), (4096, 512, 96, 0), (8192, 512, 0, 0), (8192, 512, 96, 0), (16384, 512, 0, 0), (16384, 512, 96, 0), (16384, 128, 96, 0)]
tot = collections.Counter()
for sg in range(n_seg):
    rng = np.random.default_rng(100 + sg)
    x = np.tile(tile, reps)
    x = (x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    one = run(x, 1 << 22, 512, 0)
    tot["one"] += len(one)
    for c in cases:
        lan = run(x, *c)
        tot[(c, "lost")] += missing(one, lan)
        tot[(c, "extra")] += missing(lan, one)
print(which, "capture: FCS-ok frames of one sequential lane per channel:", tot["one"])
print("| core / warm-up | boundaries | lost vs one lane | extra |\n|---|---|---|---|")
for c in cases:
    how = ('fixed grid' if not c[2] else 'in gaps (noise for %d samples)' % c[2]) + ('' if not c[3] else ', two starts, eye over %d chips' % c[3])
    print(f"| {c[0]} / {c[1]} | {how} | {tot[(c, 'lost')]} ({100.0 * tot[(c, 'lost')] / tot['one']:.2f} %) | {tot[(c, 'extra')]} |")
