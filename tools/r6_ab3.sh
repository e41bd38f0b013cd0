#!/bin/bash
# Dev tool (gpurun), round 6: parity first, then the grid of the correlator over the planes (SNOUT_CORR_BLOCKS: 0 = one wave
# per (slot, chunk) as in rounds 2-5) and the front-end events bound to the kernels' dispatches (SNOUT_EXT_LAUNCH) against
# the headline step; then the bench tests.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
line() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('   %.3f ms/step  kernel %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms',0)))" || tail -5 $1.err; }
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 1200 python3 -m pytest tests/test_wideband_gpu.py tests/test_btle_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4
for pass in 1 2; do
for cb in 0 1024 2048 4096; do
  for xl in 1 0; do
    export SNOUT_CORR_BLOCKS=$cb SNOUT_EXT_LAUNCH=$xl
    if [ $pass = 1 ]; then
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sync_${cb}_$xl -- python3 bench.py --steps 10 --no-cpu --no-others --sync > $O/sync_${cb}_$xl.log 2>&1
      echo "== corr blocks $cb ext $xl: sync (traced)"; line $O/sync_${cb}_$xl.log
      grep -h -E "btle_corr_planes" $(find $O/sync_${cb}_$xl -name "*kernel_stats.csv") | cut -d, -f9-
      rm -rf $O/sync_${cb}_$xl
    fi
    timeout 600 python3 bench.py --no-cpu --no-others --steps 20 --warmup 3 > $O/plain_${cb}_${xl}_$pass.log 2> $O/plain_${cb}_${xl}_$pass.log.err
    echo "== corr blocks $cb ext $xl: plain"; line $O/plain_${cb}_${xl}_$pass.log
  done
done
done
unset SNOUT_CORR_BLOCKS SNOUT_EXT_LAUNCH
timeout 3000 python3 -m pytest tests/test_bench_gpu.py tests/test_gather_gpu.py tests/test_scan_gpu.py -x -q -m gpu 2>&1 | tail -15
