#!/bin/bash
# Dev tool (gpurun), round 6: wideband 802.15.4 split mode (SNOUT_ZB_SPLIT = R: the channelizer's grid leaves R CUs, segment
# i's lanes run beside segment i + 1's channelizer on them): parity first, then cfg #4 / cfg #5 step time by R.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step  kernel %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0)))" || tail -5 $1.err; }
SNOUT_ZB_SPLIT=80 timeout 1500 python3 -m pytest tests/test_wideband_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4
for pass in 1 2; do
for r in 0 64 72 80 88 96; do
  export SNOUT_ZB_SPLIT=$r
  timeout 600 python3 bench.py --no-cpu --workload cfg4 --steps 20 --warmup 3 > $O/cfg4_${r}_$pass.log 2> $O/cfg4_${r}_$pass.log.err
  echo "== split $r: cfg4"; line $O/cfg4_${r}_$pass.log
  if [ $pass = 1 ]; then
    timeout 600 python3 bench.py --no-cpu --workload cfg5 --steps 20 --warmup 3 > $O/cfg5_${r}.log 2> $O/cfg5_${r}.log.err
    echo "== split $r: cfg5"; line $O/cfg5_${r}.log
  fi
done
done
export SNOUT_ZB_SPLIT=80
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace80 -- python3 bench.py --no-cpu --workload cfg4 --steps 10 --warmup 3 > $O/trace80.log 2>&1
python3 tools/timeline.py $O/trace80 --marker "pfb_spec<16" --last 6 --per 1 --skip 3 --list 8 > $O/trace80_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -30 $O/trace80_timeline.txt
