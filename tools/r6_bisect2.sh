#!/bin/bash
# Dev tool (gpurun), round 6: the trees of five round-5 commits (build/bis/<sha>, archives built in place), round 4's and
# HEAD on ONE box: btle_corr_planes one segment at a time (kernel trace kept: what runs around it), and the 8-block
# rehearsal of rank 0's load untraced.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b; mkdir -p $O
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step  kernel %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0)))"; }
for T in r4 9940954 ca86f7c a3be888 fb34e5e 3f8f012 head; do
  D=$R/build/bis/$T; [ $T = r4 ] && D=$R/build/r4tree; [ $T = head ] && D=$R
  cd $D
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sync_$T -- python3 bench.py --steps 10 --no-cpu --no-others --sync > $O/sync_$T.log 2>&1
  echo "== $T sync (traced)"; line $O/sync_$T.log
  grep -h -E "btle_corr_planes" $(find $O/sync_$T -name "*kernel_stats.csv") | cut -d, -f9-
  find $O/sync_$T -name "*agent_info.csv" -delete
  for fw in 0 8; do
      SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=$fw timeout 600 python3 bench.py --no-cpu --no-others --steps 20 --warmup 3 > $O/fw${fw}_$T.log 2>/dev/null
      echo "== $T fake world $fw"; line $O/fw${fw}_$T.log
  done
done
cd $R/build/r4tree
SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/fwtrace_r4 -- python3 bench.py --no-cpu --no-others --steps 12 --warmup 3 > $O/fwtrace_r4.log 2>&1
cd $R
SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/fwtrace_head -- python3 bench.py --no-cpu --no-others --steps 12 --warmup 3 > $O/fwtrace_head.log 2>&1
