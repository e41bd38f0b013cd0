#!/bin/bash
# Dev tool (gpurun): cfg5 step time for a few scan configurations, two runs each
R=${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; for i in 1 2; do env "$@" python $R/bench.py --workload cfg5 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   %.3f ms/step  %.1f Gsamples/s  crc_ok %d' % (d['ms_per_step'], d['value']/1e3, d['config']['decoded_crc_ok']))"; done; }
run A=1
run SNOUT_CFG5_ZPRIO=-1
run SNOUT_CFG5_BPRIO=-1
run SNOUT_CFG5_ZPRIO=-1 SNOUT_CFG5_BB=24
run SNOUT_CFG5_BB=24
run A=1
