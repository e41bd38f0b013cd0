#!/bin/bash
# Dev tool (gpurun): cfg #4 with the frame repair -- parity check, then kernel durations one segment at a time and pipelined
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=${CORE:-8192}; W=${WARM:-1024}
python3 tools/r5_repair_check.py 1 2>&1 | tail -5
for m in sync pipe pipe8; do
  f=""; r=0; [ $m = sync ] && f="--sync"; [ $m = pipe8 ] && r=8
  SNOUT_BENCH_RESERVED_CUS=$r SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W rocprofv3 --kernel-trace -d gpurun_out/r5_prof_$m -o a -- python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 $f > gpurun_out/r5_prof_$m.log 2>&1
  echo "== $m traced: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_prof_$m.log)"
  python3 tools/r5_kstats.py gpurun_out/r5_prof_$m/a_results.db zb_ | head -4
  python3 tools/r5_kstats.py gpurun_out/r5_prof_$m/a_results.db pfb
done
for r in 0 4 8 16; do
  echo "== untraced, reserved CUs $r: $(SNOUT_BENCH_RESERVED_CUS=$r SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
echo "== untraced, no repair: $(SNOUT_ZB_REPAIR=0 SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
