#!/bin/bash
# Dev tool (run through gpurun): kernel trace of `bench.py --workload $WL` (default cfg4) and a timeline summary of the last
# 10 steps: how much of the wall time some kernel runs, how much two or more run, who runs alone.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wl_tl; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload ${WL:-cfg4} --no-cpu --steps 10 --warmup 3 > $O/run.log 2>&1
tail -1 $O/run.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$O/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("snout::", "")[:28], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
# the timed region: the last 6 steps = the last 6 x 48 launches of the 40-channel channelizer
p40 = [e[0] for e in ev if e[2].startswith("void pfb_spec") or e[2].startswith("void zb_discrim") or e[2].startswith("btle_demod")]
lo, hi = p40[-11], p40[-1]          # the ten timed steps: the last front-end launch belongs to the check behind them
ev = [e for e in ev if lo <= e[0] < hi]
span = max(e[1] for e in ev) - min(e[0] for e in ev)
pts = []
for s, e, n, q, st in ev:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
busy1 = busy2 = 0; cur = 0; last = pts[0][0]
alone = collections.Counter(); live = collections.Counter()
for t, d, n in pts:
    dt = t - last
    if cur >= 1: busy1 += dt
    if cur >= 2: busy2 += dt
    if cur == 1:
        alone[[k for k, v in live.items() if v > 0][0]] += dt
    live[n] += d; cur += d; last = t
print("window %.2f ms; some kernel running %.1f %%, two or more %.1f %%, idle %.1f %%" % (span / 1e6, 100 * busy1 / span, 100 * busy2 / span, 100 * (span - busy1) / span))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q, st in ev:
    tot[n] += e - s; cnt[n] += 1
print("kernel                        calls   sum ms   avg us   alone ms")
for n, v in tot.most_common(16):
    print("%-28s %6d %8.2f %8.1f %8.2f" % (n, cnt[n], v / 1e6, v / cnt[n] / 1e3, alone[n] / 1e6))
print("queues used:", sorted(set(e[3] for e in ev)), "streams:", len(set(e[4] for e in ev)))
PY
rm -f $O/*/*kernel_trace.csv $O/*/*/*kernel_trace.csv $O/*/*agent_info.csv
