#!/bin/bash
# Dev tool: build variants of libsnout_rx.so that differ in the compile-time switches of pfb.hip / pfb_mfma.hip, for A/B
# timing in one gpurun call (tools/pfb_ab.py loads them through SNOUT_RX_LIB).
#   tools/pfb_variants.sh name1:"-DFOO -DBAR=1" name2:"" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/snout_amd/csrc"
OUT="$ROOT/build/variants"; mkdir -p "$OUT/obj"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-function"
for f in btle.hip zigbee.hip membench.hip records.hip formats.cpp snout_rx.cpp; do
  o="$OUT/obj/${f%.*}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ]; then /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$o" & fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs -c pfb.hip -o "$OUT/obj/pfb_$name.o" &&
    /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs -c pfb_mfma.hip -o "$OUT/obj/pfb_mfma_$name.o" &&
    /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs -c pfb_spec.hip -o "$OUT/obj/pfb_spec_$name.o" &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libsnout_rx_$name.so" "$OUT/obj/pfb_$name.o" "$OUT/obj/pfb_mfma_$name.o" "$OUT/obj/pfb_spec_$name.o" \
      "$OUT/obj/btle.o" "$OUT/obj/zigbee.o" "$OUT/obj/membench.o" "$OUT/obj/records.o" "$OUT/obj/formats.o" "$OUT/obj/snout_rx.o" && echo "built $name [$defs]" ) &
done
wait
