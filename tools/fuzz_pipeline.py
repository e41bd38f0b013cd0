"""Randomised check of the pipelined entry points (developer tool): segments of random sizes through
submit / collect, up to three in flight, must give exactly the records of one-at-a-time process()
calls -- across the small/large-segment scheduling modes and the result-slot / work-set rotation.
Wideband handles also go through submit_batch(): random runs of equal-length segments as one
submission, with random min indices, against the per-segment process() results."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx

FIELDS = ("sample_index", "proto", "channel", "len", "crc_ok", "lqi", "pdu_type", "flags", "aux")


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in FIELDS) and np.array_equal(a["bytes"], b["bytes"])


def main(budget_s):
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    t_end = time.time() + budget_s
    kinds = ("btle", "zigbee", "btle40", "zigbee16")
    done = {k: 0 for k in kinds}
    while time.time() < t_end:
        kind = str(rng.choice(kinds))
        seed = int(rng.integers(1 << 30))
        if kind == "btle":
            big, _ = synth.btle_capture(1 << 21, channel=37, seed=seed, mean_gap=3000.0)
            kw, unit, lo = dict(proto=0, channel=37), 1, 5
        elif kind == "zigbee":
            big, _ = synth.zigbee_capture(1 << 20, seed=seed, mean_gap=6000.0)
            kw, unit, lo = dict(proto=1, channel=11), 1, 9
        elif kind == "btle40":
            big, _ = synth.wideband_capture(0, 40 * 50000, seed=seed % 1000, bins=[1, 9, 30], mean_gap=4000.0)
            kw, unit, lo = dict(proto=0, n_channels=40), 1, 700
        else:
            big, _ = synth.wideband_capture(1, 16 * 60000, seed=seed % 1000, bins=[2, 11], mean_gap=6000.0, max_len=40)
            kw, unit, lo = dict(proto=1, n_channels=16), 1, 300
        t = torch.from_numpy(big.view(np.float32)).cuda()
        # BTLE narrowband: repeat so that some segments exceed the 2^26-sample scheduling threshold
        if kind == "btle" and rng.random() < 0.3:
            t = t.repeat(40)
        n_tot = t.numel() // 2
        segs = []
        for _ in range(int(rng.integers(3, 9))):
            n = int(rng.integers(lo, n_tot))
            a = int(rng.integers(0, n_tot - n + 1))
            segs.append((a, n, int(rng.integers(0, 1 << 40))))
        with SnoutRx(**kw) as rx:
            want = [rx.process(t[2 * a:2 * (a + n)], first_sample_index=f) for a, n, f in segs]
            got, inflight = [], 0
            for a, n, f in segs:
                if inflight == 3:
                    got.append(rx.collect()); inflight -= 1
                rx.submit(t[2 * a:2 * (a + n)], first_sample_index=f); inflight += 1
                if rng.random() < 0.3 and inflight:
                    got.append(rx.collect()); inflight -= 1
            while inflight:
                got.append(rx.collect()); inflight -= 1
        if len(got) != len(want) or not all(same(g, w) for g, w in zip(got, want)):
            print("MISMATCH:", kind, "seed", seed, segs, flush=True)
            return 1
        done[kind] += 1
        if kind in ("btle40", "zigbee16"):
            # batches: equal lengths; up to two batches in flight
            B = int(rng.integers(2, 9)) if rng.random() < 0.6 else int(rng.integers(9, 65))       # up to 64 segments per submission (round 4)
            n = int(rng.integers(lo * 4, n_tot // 2))
            subs = []
            for _ in range(int(rng.integers(2, 5))):
                cnt = int(rng.integers(1, B + 1))
                items = []
                for _ in range(cnt):
                    a = int(rng.integers(0, n_tot - n + 1))
                    f = int(rng.integers(0, 1 << 40))
                    m = f + int(rng.integers(0, 2 * n // (20 if kind == "btle40" else 8))) if rng.random() < 0.5 else 0
                    items.append((a, f, m))
                subs.append(items)
            with SnoutRx(**kw) as one, SnoutRx(batch_segments=B, **kw) as rx:
                want = []
                for items in subs:
                    parts = []
                    for a, f, m in items:
                        r = one.process(t[2 * a:2 * (a + n)], first_sample_index=f)
                        parts.append(r[r["sample_index"] >= m])
                    want.append(np.concatenate(parts))
                got, inflight = [], 0
                for items in subs:
                    if inflight == 2:
                        got.append(rx.collect()); inflight -= 1
                    rx.submit_batch([t[2 * a:2 * (a + n)] for a, _, _ in items], [f for _, f, _ in items], [m for _, _, m in items])
                    inflight += 1
                while inflight:
                    got.append(rx.collect()); inflight -= 1
            if len(got) != len(want) or not all(same(g, w) for g, w in zip(got, want)):
                print("BATCH MISMATCH:", kind, "seed", seed, "n", n, subs, flush=True)
                return 1
            done[kind + "_batches"] = done.get(kind + "_batches", 0) + 1
    print("pipeline fuzz ok:", done, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0))
