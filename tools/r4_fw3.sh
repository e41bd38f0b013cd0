#!/bin/bash
# Dev tool (gpurun): the headline workload with rank 0's 8-rank load rehearsed, by how the download is done
export SNOUT_BENCH_NCCL1=1
R=${GRAFT_REPO_ROOT:-.}
WL=${WL:-cfg3}
run() { echo "== $*"; for i in 1 2; do env "$@" python $R/bench.py --no-cpu --steps 20 --warmup 3 --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('   %.3f ms/step  %.1f Gsamples/s  kernel %.3f  records on rank 0 %s' % (d['ms_per_step'], d['value']/1e3, d['roofline'].get('kernel_ms',0), c.get('records_on_rank0_last_step', c.get('records_on_rank0'))))"; done; }
run A=1
run SNOUT_BENCH_FAKE_WORLD=8
run SNOUT_BENCH_FAKE_WORLD=8 SNOUT_GATHER_COPY_WGS=4
run SNOUT_BENCH_FAKE_WORLD=8 SNOUT_GATHER_COPY_WGS=16 SNOUT_BENCH_RESERVED_CUS=16
run SNOUT_BENCH_FAKE_WORLD=8 SNOUT_GATHER_COPY_WGS=0
run SNOUT_BENCH_FAKE_WORLD=8 SNOUT_BENCH_RESERVED_CUS=0
