#!/bin/bash
# Dev tool (run through gpurun): kernel trace of one bench workload and its timeline summary.
#   WL=cfg5 MARK='pfb_spec<40' PER=2 STEPS=6 LIST=20 bash tools/r4_tl.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; WL=${WL:-cfg5}; O=$R/gpurun_out/tl_$WL; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $WL --no-cpu --steps ${STEPS:-6} --warmup 3 "$@" > $O/run.log 2>&1
tail -1 $O/run.log | cut -c1-400
python3 $R/tools/timeline.py $O --marker "${MARK:-pfb_spec<40}" --last ${STEPS:-6} --per ${PER:-1} --list ${LIST:-0} > $O/timeline.txt 2>&1
head -40 $O/timeline.txt
rm -f $O/*/*kernel_trace.csv $O/*/*/*kernel_trace.csv $O/*/*agent_info.csv
