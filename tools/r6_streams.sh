#!/bin/bash
# Dev tool (gpurun), round 6: the number of tail streams a handle CREATES (SNOUT_TAIL_STREAMS = 1 | 3) against
# btle_corr_planes one segment at a time, the pipelined headline, the 8-block rehearsal of rank 0's load, cfg #4 and cfg #5
# -> profiles/r6_streams.md.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step  kernel %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0)))"; }
for ts in 1 3 1 3; do
  export SNOUT_TAIL_STREAMS=$ts
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sync_$ts -- python3 bench.py --steps 10 --no-cpu --no-others --sync > $O/sync_$ts.log 2>&1
  echo "== tails $ts: cfg3 sync (traced)"; line $O/sync_$ts.log
  grep -h -E "btle_corr_planes" $(find $O/sync_$ts -name "*kernel_stats.csv") | cut -d, -f9-
  rm -rf $O/sync_$ts
  for fw in 0 8; do
      SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=$fw timeout 600 python3 bench.py --no-cpu --no-others --steps 20 --warmup 3 > $O/fw${fw}_$ts.log 2>/dev/null
      echo "== tails $ts: cfg3 fake world $fw"; line $O/fw${fw}_$ts.log
  done
  timeout 600 python3 bench.py --no-cpu --no-others --steps 20 --warmup 3 > $O/plain_$ts.log 2>/dev/null
  echo "== tails $ts: cfg3 plain"; line $O/plain_$ts.log
  for w in cfg4 cfg5; do
    timeout 600 python3 bench.py --no-cpu --workload $w --steps 20 --warmup 3 > $O/${w}_$ts.log 2>/dev/null
    echo "== tails $ts: $w"; line $O/${w}_$ts.log
  done
  SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=8 timeout 600 python3 bench.py --no-cpu --workload cfg5 --steps 20 --warmup 3 > $O/cfg5fw_$ts.log 2>/dev/null
  echo "== tails $ts: cfg5 fake world 8"; line $O/cfg5fw_$ts.log
done
