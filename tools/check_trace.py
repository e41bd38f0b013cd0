#!/usr/bin/env python3
"""Is a rocprofv3 kernel trace a trace of the steady state?  (round 5's cfg #5 profile had the in-run
one-lane check inside it -- `zb_mm` max 805 ms against a median of a few ms.)

    python3 tools/check_trace.py <dir with *kernel_trace.csv> [kernel name fragment ...]

For every kernel whose name contains one of the fragments (default: the kernel with the largest total time) prints calls,
median, average, max; exits 1 -- and writes NOT_STEADY into <dir>/steady.txt -- if some max exceeds 10 x the median
(tools/collect_profiles_r6.py refuses such a directory)."""
import collections
import csv
import glob
import os
import statistics
import sys

d = sys.argv[1]
frags = sys.argv[2:]
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
if not fs:
    raise SystemExit("no kernel trace under " + d)
dur = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = [k for k in dur if any(f in k for f in frags)] if frags else [max(dur, key=lambda k: sum(dur[k]))]
bad = False
lines = []
for k in sorted(names, key=lambda k: -sum(dur[k])):
    v = dur[k]
    med, mx = statistics.median(v), max(v)
    flag = mx > 10 * med and len(v) >= 4
    bad |= flag
    lines.append("%-44s calls %5d  median %10.1f us  avg %10.1f us  max %10.1f us%s"
                 % (k[:44], len(v), med / 1e3, sum(v) / len(v) / 1e3, mx / 1e3, "   <- max > 10 x median: not the steady state" if flag else ""))
print("\n".join(lines))
open(os.path.join(d, "steady.txt"), "w").write(("NOT_STEADY\n" if bad else "STEADY\n") + "\n".join(lines) + "\n")
sys.exit(1 if bad else 0)
