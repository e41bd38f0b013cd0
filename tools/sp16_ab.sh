#!/bin/bash
# Dev tool: A/B of the M = 16 pfb_spec layouts (SNOUT_SP16_LAYOUT, pfb_spec.hip) in one gpurun call.  Build first:
#   tools/pfb_variants.sh l0:"-DSNOUT_SP16_LAYOUT=0" l1:"-DSNOUT_SP16_LAYOUT=1" l2:"-DSNOUT_SP16_LAYOUT=2"
cd "$(dirname "$0")/.."
N=${1:-3.2e8}
for v in ${VARIANTS:-l0 l1 l2 l0 l1 l2}; do
  echo "== $v"
  SNOUT_RX_LIB=build/variants/libsnout_rx_$v.so timeout 300 python tools/pfb_ab.py --child --proto 1 --samples $N 2>&1 | tail -1
done
