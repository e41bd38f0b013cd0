#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: how much of a window some kernel runs, how much two or more run,
who runs alone, and (--list N) the launches of the last N ms in order with the idle gaps before them.

    python3 tools/timeline.py <dir with *kernel_trace.csv> --marker 'pfb_spec<40' --last 6 [--per 2] [--list 12]

The window starts at the (last x per)-th from last launch of the marker kernel (`per` launches of it per step) and ends
with the last kernel of the trace (pass --until-marker to end at the marker's last launch instead, --skip N to leave out
the last N marker launches first)."""
import argparse
import collections
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--marker", default="pfb_spec<40")
ap.add_argument("--last", type=int, default=6, help="steps in the window")
ap.add_argument("--per", type=int, default=1, help="marker launches per step")
ap.add_argument("--until-marker", action="store_true")
ap.add_argument("--skip", type=int, default=0, help="leave out the last N marker launches (what a bench runs after its timed loop); implies --until-marker")
ap.add_argument("--list", type=float, default=0.0, help="list the launches of the last N ms of the window")
a = ap.parse_args()
f = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("snout::", "").replace("void ", "")[:30],
       r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
mk = [e[0] for e in ev if a.marker in e[2]]
if a.skip:
    mk = mk[:len(mk) - a.skip]
    a.until_marker = True
need = a.last * a.per
if len(mk) < need + 1:
    raise SystemExit("only %d launches of %r in the trace" % (len(mk), a.marker))
lo = mk[-need - (1 if a.until_marker else 0)] if not a.until_marker else mk[-need - 1]
hi = mk[-1] if a.until_marker else max(e[1] for e in ev) + 1
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
print("window %.2f ms = %d steps of %.3f ms; some kernel running %.1f %%, two or more %.1f %%, idle %.1f %%"
      % (span / 1e6, a.last, span / 1e6 / a.last, 100 * busy1 / span, 100 * busy2 / span, 100 * (span - busy1) / span))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q, st in ev:
    tot[n] += e - s; cnt[n] += 1
print("kernel                          calls   sum ms  ms/step   avg us   alone ms")
for n, v in tot.most_common(22):
    print("%-30s %6d %8.2f %8.3f %8.1f %8.2f" % (n, cnt[n], v / 1e6, v / 1e6 / a.last, v / cnt[n] / 1e3, alone[n] / 1e6))
print("queues used:", sorted(set(e[3] for e in ev)), "streams:", len(set(e[4] for e in ev)))
if a.list:
    t_end = max(e[1] for e in ev)
    t0 = t_end - int(a.list * 1e6)
    prev_end = None
    print("\nlaunches of the last %.1f ms (start us, duration us, queue, kernel; gap = GPU idle before it)" % a.list)
    running_end = 0
    for s, e, n, q, st in ev:
        if s < t0:
            running_end = max(running_end, e)
            continue
        gap = s - running_end if running_end and s > running_end else 0
        print("%9.1f %8.1f  q%-3s %-30s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, ("   <- idle %.1f us" % (gap / 1e3)) if gap > 3000 else ""))
        running_end = max(running_end, e)
