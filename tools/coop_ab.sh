#!/bin/bash
# Dev tool: A/B of zb_walk's cooperative payload rounds (frames per round = SNOUT_ZB_COOP_GROUPS 1 / 2 / 4) in one gpurun call:
# zigbee.hip compiled three times into build/variants/libsnout_rx_coop{1,2,4}.so; parity tests first, then the kernel's time.
cd "$(dirname "$0")/.."
for G in ${GROUPS_LIST:-4 2 1}; do
  echo "== coop groups $G"
  SNOUT_RX_LIB=build/variants/libsnout_rx_coop$G.so python -m pytest tests/test_zigbee_gpu.py tests/test_wideband_gpu.py -m gpu -x -q 2>&1 | tail -1
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pz$G && SNOUT_RX_LIB=$GRAFT_REPO_ROOT/build/variants/libsnout_rx_coop$G.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pz$G -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${WL:-zigbee1} --steps 10 --no-cpu --sync > /dev/null 2>&1
   python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pz$G/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "zb_walk" in r["Name"]: print("zb_walk", r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  )
done
