#!/bin/bash
# Dev tool (gpurun): cfg #4 step time against the number of channelizer ranges (SNOUT_PFB_BLOCKS: workgroups of the launch; more
# than one per CU = the hardware hands the next range to whichever CU is free) with the frame repair running beside it
C=${CORE:-8192}; W=${WARM:-1024}
for rep in 1 2; do
for b in 256 512 768 1024 1536 2048; do
  echo "== core $C/$W blocks $b prio ${PRIO:-3}: $(SNOUT_PFB_BLOCKS=$b SNOUT_ZB_TAIL_PRIO=${PRIO:-3} SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
done
for b in 256 768 1024; do
  echo "== no repair blocks $b: $(SNOUT_ZB_REPAIR=0 SNOUT_PFB_BLOCKS=$b SNOUT_BENCH_ZB_CORE=$C SNOUT_BENCH_ZB_WARMUP=$W python3 bench.py --no-cpu --steps 10 --warmup 3 --workload cfg4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
