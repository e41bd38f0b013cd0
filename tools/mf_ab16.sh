#!/bin/bash
# Dev tool: A/B of the M = 16 channelizer kernels (802.15.4 wideband path) in one gpurun call.
cd "$(dirname "$0")/.."
N=${1:-3.2e8}
for impl in ${IMPLS:-spec valu spec valu}; do
  echo "== $impl"
  SNOUT_PFB_IMPL=$impl timeout 300 python tools/pfb_ab.py --child --proto 1 --samples $N 2>&1 | tail -1
done
