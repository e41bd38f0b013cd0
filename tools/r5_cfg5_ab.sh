#!/bin/bash
# Dev tool (gpurun): cfg #5 on one GPU by handles per scan / segments per submission / priorities
run() { echo "== $*: $(env "$@" python3 bench.py --workload cfg5 --steps 8 --warmup 3 --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
run A=1
run SNOUT_CFG5_HZ=2
run SNOUT_CFG5_HZ=3
run SNOUT_CFG5_HZ=2 SNOUT_CFG5_HB=2
run SNOUT_CFG5_HZ=2 SNOUT_ZB_TAIL_PRIO=0
run SNOUT_CFG5_HZ=3 SNOUT_ZB_TAIL_PRIO=0
run SNOUT_CFG5_BZ=10 SNOUT_CFG5_HZ=2
run SNOUT_CFG5_ZB_CORE=4096 SNOUT_CFG5_ZB_WARMUP=512
run SNOUT_CFG5_ZB_CORE=4096 SNOUT_CFG5_ZB_WARMUP=512 SNOUT_ZB_REPAIR=0
