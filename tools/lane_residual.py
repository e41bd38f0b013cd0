"""Dev tool (CPU, the oracle = what the GPU computes bit for bit): the residual of the 802.15.4 lane decomposition
against ONE sequential lane, by noise level and lane warm-up, split by mechanism.  Writes profiles/r3_lane_residual.md.

    python tools/lane_residual.py [tile repeats, default 6]

Per cell: frames of the one-lane run | lost / extra with the default lanes (core 2048) | of the lanes' frames, how many
carry SNOUT_PKT_ZB_SEAM_DISAGREED, and how many of the EXTRA ones do | lost / extra with the oracle's analysis switch
ORACLE_ZB_EXPERIMENT_RESTART (a lane's sink that would start inside a kept frame starts behind it instead: the part of
the residual that is the sink's history, not the timing loop's)."""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(sigma, warm, reps, restart):
    if restart:
        os.environ["ORACLE_ZB_EXPERIMENT_RESTART"] = "1"
    else:
        os.environ.pop("ORACLE_ZB_EXPERIMENT_RESTART", None)
    from snout_amd import synth
    from oracle import oracle_py
    oracle_py.set_threads(os.cpu_count())
    tz, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
    rng = np.random.default_rng(5)
    x = np.tile(tz, reps)
    x = (x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    one = oracle_py.wideband_segment(x, proto=1, core=1 << 22, warmup=warm)
    lan = oracle_py.wideband_segment(x, proto=1, core=2048, warmup=warm)
    key = lambda a: [(int(c), bytes(b[:l]), int(s)) for c, s, l, b in zip(a["channel"], a["sample_index"], a["len"], a["bytes"])]
    A, B = key(one), key(lan)

    def missing(P, Q):
        d = collections.defaultdict(list)
        for c, b, s in Q:
            d[(c, b)].append(s)
        return [i for i, (c, b, s) in enumerate(P) if not any(abs(s - u) <= 8 for u in d.get((c, b), []))]
    lost, extra = missing(A, B), missing(B, A)
    flagged = (lan["flags"] & 4) != 0
    return dict(one=len(A), lanes=len(B), lost=len(lost), extra=len(extra), flagged=int(flagged.sum()),
                extra_flagged=int(sum(bool(flagged[i]) for i in extra)), one_flagged=int(((one["flags"] & 4) != 0).sum()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--cell":
        sigma, warm, reps, restart = float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        print(repr(run(sigma, warm, reps, restart)))
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rows = []
    for sigma in (0.02, 0.05, 0.1):
        for warm in (256, 512, 1024):
            cell = []
            for restart in (0, 1):          # a child per cell: the switch is read through getenv inside the oracle
                out = subprocess.run([sys.executable, __file__, "--cell", str(sigma), str(warm), str(reps), str(restart)],
                                     capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
                cell.append(eval(out))
            a, b = cell
            rows.append((sigma, warm, a, b))
            print(sigma, warm, a, b, flush=True)
    with open(os.path.join(ROOT, "profiles", "r3_lane_residual.md"), "w") as f:
        f.write("# 802.15.4: default lanes (core 2048) against one sequential lane, by noise and warm-up (oracle = GPU, bit for bit)\n\n")
        f.write("16-channel wideband capture (8 busy bins, frames up to 100 bytes), %d x 2^21 input samples, AWGN sigma per component on top\n" % reps)
        f.write("of the channelizer's leakage; `tools/lane_residual.py`.  lost / extra: frames of the one-lane run the lanes do not report / report in\n")
        f.write("addition (whole frames, identical bytes otherwise).  flagged: lanes' frames carrying SNOUT_PKT_ZB_SEAM_DISAGREED.\n")
        f.write("restart: the same with the oracle's analysis switch (a sink that would start inside a kept frame starts behind it).\n\n")
        f.write("| sigma | warm-up | frames (one lane) | lost | extra | flagged (of lanes' frames) | extra that are flagged | one-lane flagged | lost / extra with restart |\n|---|---|---|---|---|---|---|---|---|\n")
        for sigma, warm, a, b in rows:
            f.write("| %.2f | %d | %d | %d | %d | %d (%.1f %%) | %d of %d | %d | %d / %d |\n" % (
                sigma, warm, a["one"], a["lost"], a["extra"], a["flagged"], 100.0 * a["flagged"] / max(1, a["lanes"]),
                a["extra_flagged"], a["extra"], a["one_flagged"], b["lost"], b["extra"]))
