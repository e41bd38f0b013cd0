#!/bin/bash
# Dev tool (gpurun), round 6: tail streams 1 | 3 for the 802.15.4 workloads and cfg #5 (+ its 8-block rehearsal); then the
# default bench line with the multi-tile captures.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step  kernel %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0)))" || tail -5 $1.err; }
for ts in 1 3 1 3; do
  export SNOUT_TAIL_STREAMS=$ts
  for w in zigbee1 cfg4 cfg5; do
    timeout 600 python3 bench.py --no-cpu --workload $w --steps 20 --warmup 3 > $O/${w}_$ts.log 2> $O/${w}_$ts.log.err
    echo "== tails $ts: $w"; line $O/${w}_$ts.log
  done
  SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=8 timeout 600 python3 bench.py --no-cpu --workload cfg5 --steps 20 --warmup 3 > $O/cfg5fw_$ts.log 2> $O/cfg5fw_$ts.log.err
  echo "== tails $ts: cfg5 fake world 8"; line $O/cfg5fw_$ts.log
done
unset SNOUT_TAIL_STREAMS
timeout 1500 python3 bench.py > $O/bench.log 2> $O/bench.err
tail -c 6000 $O/bench.log; tail -5 $O/bench.err
