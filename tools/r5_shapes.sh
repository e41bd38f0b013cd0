#!/bin/bash
# Dev tool (gpurun): what the 802.15.4 lane shape and the frame repair cost on cfg #4 (3.2e8 samples, one segment)
run() { echo "== $WL $*"; env "$@" python bench.py --no-cpu --steps 10 --warmup 3 --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('   %.3f ms/step  %.1f Gsamples/s  crc_ok %s' % (d['ms_per_step'], d['value']/1e3, c.get('decoded_crc_ok_per_gpu', c.get('decoded_crc_ok'))))"; }
WL=${1:-cfg4}
run SNOUT_ZB_REPAIR=0 SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=512
run SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=512
run SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=1024
run SNOUT_BENCH_ZB_CORE=6144 SNOUT_BENCH_ZB_WARMUP=1024
run SNOUT_ZB_REPAIR=0 SNOUT_BENCH_ZB_CORE=8192 SNOUT_BENCH_ZB_WARMUP=1024
run SNOUT_BENCH_ZB_CORE=8192 SNOUT_BENCH_ZB_WARMUP=1024
run SNOUT_BENCH_ZB_CORE=8192 SNOUT_BENCH_ZB_WARMUP=2048
run SNOUT_BENCH_ZB_CORE=12288 SNOUT_BENCH_ZB_WARMUP=2048
