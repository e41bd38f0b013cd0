#!/bin/bash
# Dev tool (gpurun), round 6: (1) btle_corr_planes / the headline step on round 4's tree (build/r4tree, an archive of
# bddf6d5 built in place) against HEAD on ONE box, one segment at a time and pipelined; (2) rank 0's load at N = 8
# (SNOUT_BENCH_FAKE_WORLD=8) on both trees, and a kernel + HIP-API trace of HEAD's rehearsal.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a; mkdir -p $O
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step  kernel %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0)))"; }
for T in r4 head r4 head; do
  D=$R; [ $T = r4 ] && D=$R/build/r4tree
  cd $D
  for m in sync pipe; do
    f=""; [ $m = sync ] && f="--sync"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${m}_$T -- python3 bench.py --steps 20 --no-cpu --no-others $f > $O/${m}_$T.log 2>&1
    echo "== $T $m (traced)"; line $O/${m}_$T.log
    grep -h -E "btle_corr_planes|pfb_spec|btle_flatten|btle_decode|btle_resolve|btle_emit" $(find $O/${m}_$T -name "*kernel_stats.csv") | cut -c1-160
  done
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
done
export SNOUT_BENCH_NCCL1=1
for T in r4 head; do
  D=$R; [ $T = r4 ] && D=$R/build/r4tree
  cd $D
  for fw in 0 8; do
    for i in 1 2; do
      SNOUT_BENCH_FAKE_WORLD=$fw timeout 600 python3 bench.py --no-cpu --no-others --steps 20 --warmup 3 > $O/fw${fw}_$T.log 2>/dev/null
      echo "== $T fake world $fw"; line $O/fw${fw}_$T.log
    done
  done
done
cd $R
SNOUT_BENCH_FAKE_WORLD=8 timeout 600 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d $O/fwtrace -- python3 bench.py --no-cpu --no-others --steps 12 --warmup 3 > $O/fwtrace.log 2>&1
ls -la $(find $O/fwtrace -name "*.csv") | head
# keep the traces small enough to travel back: the last 40000 lines of each
for f in $(find $O/fwtrace -name "*_trace.csv"); do (head -1 $f; tail -n 40000 $f) > $f.tail; rm $f; done
