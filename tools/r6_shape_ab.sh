#!/bin/bash
# Dev tool (gpurun), round 6: the default lane shape (6144 / 3072) against round 5's (6144 / 1024) through bench.py itself,
# alternating, cfg #4 and cfg #5.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step' % d['ms_per_step'])"; }
for pass in 1 2 3; do
  for w in 1024 2048 3072; do
    SNOUT_BENCH_ZB_CORE=6144 SNOUT_BENCH_ZB_WARMUP=$w timeout 600 python3 bench.py --no-cpu --workload cfg4 --steps 20 --warmup 3 > $O/cfg4_$w.log 2>/dev/null
    echo "== cfg4 6144 / $w"; line $O/cfg4_$w.log
    SNOUT_CFG5_ZB_CORE=6144 SNOUT_CFG5_ZB_WARMUP=$w timeout 600 python3 bench.py --no-cpu --workload cfg5 --steps 20 --warmup 3 > $O/cfg5_$w.log 2>/dev/null
    echo "== cfg5 6144 / $w"; line $O/cfg5_$w.log
  done
done
