#!/bin/bash
# Dev tool: cfg #5 on one GPU: segments per submission (BTLE, 802.15.4) and handles per scan.
cd "$(dirname "$0")/.."
for combo in ${COMBOS:-"4 4 1 2" "8 8 1 2" "4 8 1 2" "8 8 1 1" "8 8 2 2" "4 4 1 2"}; do
  set -- $combo
  echo "== BB=$1 BZ=$2 HB=$3 HZ=$4"
  SNOUT_CFG5_BB=$1 SNOUT_CFG5_BZ=$2 SNOUT_CFG5_HB=$3 SNOUT_CFG5_HZ=$4 timeout 300 python bench.py --workload cfg5 --steps 6 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
