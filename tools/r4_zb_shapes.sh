#!/bin/bash
# Dev tool (gpurun): what the 802.15.4 lane shape costs on cfg #4 (3.2e8 samples, one segment) and cfg #5
run() { echo "== $WL $*"; env "$@" python bench.py --no-cpu --steps 10 --warmup 3 --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('   %.3f ms/step  %.1f Gsamples/s  crc_ok %s' % (d['ms_per_step'], d['value']/1e3, c.get('decoded_crc_ok_per_gpu', c.get('decoded_crc_ok'))))"; }
WL=cfg4
run A=1
run SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=512
run SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=2048
run SNOUT_BENCH_ZB_CORE=8192 SNOUT_BENCH_ZB_WARMUP=4096
run SNOUT_BENCH_ZB_CORE=16384 SNOUT_BENCH_ZB_WARMUP=8192
WL=cfg5
run A=1
run SNOUT_CFG5_ZB_CORE=4096 SNOUT_CFG5_ZB_WARMUP=2048
run SNOUT_CFG5_ZB_CORE=8192 SNOUT_CFG5_ZB_WARMUP=4096
run SNOUT_CFG5_ZB_CORE=16384 SNOUT_CFG5_ZB_WARMUP=8192
