#!/bin/bash
# Dev tool (run through gpurun): kernel trace + stats of tools/cfg5.py (both wideband scans in 2^24-sample segments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_cfg5; mkdir -p $O
HB=1 HZ=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/cfg5.py > $O/run.log 2>&1
tail -3 $O/run.log
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.1f ms over the whole process" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-70s calls %6s total %8.2f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
