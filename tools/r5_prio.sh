#!/bin/bash
# Dev tool (gpurun): cfg #4 step time under tail priorities / lane shapes
C=${CORE:-8192}; W=${WARM:-1024}
for rep in 1 2; do
for p in 3 0 1 2; do
  echo "== core $C/$W tail_prio $p: $(SNOUT_ZB_TAIL_PRIO=$p SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
done
for s in "6144 1024" "12288 1024" "4096 512"; do set -- $s
  echo "== core $1/$2 tail_prio 0: $(SNOUT_ZB_TAIL_PRIO=0 SNOUT_BENCH_ZB_CORE=$1 SNOUT_BENCH_ZB_WARMUP=$2 python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
echo "== no repair 8192/1024 prio 0: $(SNOUT_ZB_REPAIR=0 SNOUT_ZB_TAIL_PRIO=0 SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
echo "== no repair 4096/512 prio 0: $(SNOUT_ZB_REPAIR=0 SNOUT_ZB_TAIL_PRIO=0 SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=512 python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
echo "== no repair 4096/512 prio 3: $(SNOUT_ZB_REPAIR=0 SNOUT_ZB_TAIL_PRIO=3 SNOUT_BENCH_ZB_CORE=4096 SNOUT_BENCH_ZB_WARMUP=512 python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
