#!/bin/bash
# Dev tool (gpurun): the serial kernels of the 802.15.4 chain against the lane length, one segment at a time (cfg #4):
# is zb_mm's time per chip a latency (one wave per SIMD would halve it) or an issue bound?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for C in 4096 6144 8192 10240 12288 16384; do
  SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=1024 rocprofv3 --kernel-trace -d gpurun_out/ll_$C -o a -- python3 bench.py --no-cpu --steps 6 --warmup 2 --workload cfg4 --sync > gpurun_out/ll_$C.log 2>&1
  echo "== core $C (waves of zb_mm: $((640000000 / C / 64))): $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ll_$C.log)"
  python3 tools/r5_kstats.py gpurun_out/ll_$C/a_results.db zb_ | head -3
  rm -rf gpurun_out/ll_$C
done
