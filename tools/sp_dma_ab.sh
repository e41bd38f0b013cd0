#!/bin/bash
# Dev tool: A/B of pfb_spec's input staging (LDS-DMA by the FFT waves against registers on the FIR waves) in one gpurun call.
#   tools/pfb_variants.sh dma1:"" dma0:"-DSNOUT_SP_DMA=0"
cd "$(dirname "$0")/.."
for proto in 0 1; do
  N=$([ $proto = 0 ] && echo 8e8 || echo 3.2e8)
  for v in ${VARIANTS:-dma1 dma0 dma1 dma0}; do
    echo "== proto $proto $v"
    SNOUT_RX_LIB=build/variants/libsnout_rx_$v.so timeout 300 python tools/pfb_ab.py --child --proto $proto --samples $N 2>&1 | tail -1
  done
done
