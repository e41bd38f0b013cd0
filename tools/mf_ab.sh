#!/bin/bash
# Dev tool: A/B of the two M = 40 channelizer kernels in one gpurun call (same library, SNOUT_PFB_IMPL).
#   tools/mf_ab.sh [samples]
cd "$(dirname "$0")/.."
N=${1:-8e8}
for impl in mfma valu mfma valu; do
  echo "== $impl"
  SNOUT_PFB_IMPL=$impl timeout 300 python tools/pfb_ab.py --child --proto 0 --samples $N 2>&1 | tail -2
done
