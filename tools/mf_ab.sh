#!/bin/bash
# Dev tool: A/B of the M = 40 channelizer kernels in one gpurun call.
#   IMPLS="spec valu lib:prio0" tools/mf_ab.sh [samples]      name = SNOUT_PFB_IMPL value, or lib:<variant of tools/pfb_variants.sh>
cd "$(dirname "$0")/.."
N=${1:-8e8}
for impl in ${IMPLS:-spec valu spec valu}; do
  echo "== $impl"
  if [[ $impl == lib:* ]]; then
    SNOUT_RX_LIB=build/variants/libsnout_rx_${impl#lib:}.so timeout 300 python tools/pfb_ab.py --child --proto 0 --samples $N 2>&1 | tail -1
  else
    SNOUT_PFB_IMPL=$impl timeout 300 python tools/pfb_ab.py --child --proto 0 --samples $N 2>&1 | tail -1
  fi
done
