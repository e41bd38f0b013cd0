#!/bin/bash
# Dev tool (gpurun): rank 0's load at N = 8 rehearsed on one GPU (SNOUT_BENCH_FAKE_WORLD=8 on the RCCL backend at world 1):
# the headline workload and cfg #5, without / with the rehearsal, with / without CUs reserved for the exchange.
export SNOUT_BENCH_NCCL1=1
run() { echo "== $*"; for i in 1 2; do env "$@" python bench.py --no-cpu --steps 20 --warmup 3 --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('   %.3f ms/step  %.1f Gsamples/s  kernel %.3f  records on rank 0 %s' % (d['ms_per_step'], d['value']/1e3, d['roofline'].get('kernel_ms',0), c.get('records_on_rank0_last_step', c.get('records_on_rank0'))))"; done; }
for WL in cfg3 cfg5; do
  echo "#### $WL"
  run A=1
  run SNOUT_BENCH_FAKE_WORLD=8
  run SNOUT_BENCH_FAKE_WORLD=8 SNOUT_BENCH_RESERVED_CUS=0
  run SNOUT_BENCH_FAKE_WORLD=8 SNOUT_BENCH_RESERVED_CUS=16
  run SNOUT_BENCH_RESERVED_CUS=8
done
