#!/bin/bash
# Dev tool: build variants of the library that differ in the compile-time switches of the channelizer files (pfb_spec.hip;
# with AB=1 also pfb.hip / pfb_mfma.hip, -DSNOUT_AB_KERNELS), for A/B timing in one gpurun call (tools/pfb_ab.py loads them
# through SNOUT_RX_LIB).  Every variant's ISA is checked for uses of hand-issued LDS reads before their wait
# (tools/check_lds_asm.py) before it is offered for timing.
#   [AB=1] tools/pfb_variants.sh name1:"-DFOO -DBAR=1" name2:"" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/snout_amd/csrc"
OUT="$ROOT/build/variants"; mkdir -p "$OUT/obj"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-function"
[ -n "$AB" ] && FLAGS="$FLAGS -DSNOUT_AB_KERNELS"
SFX=${AB:+_ab}
for f in btle.hip zigbee.hip membench.hip records.hip formats.cpp snout_rx.cpp pfb_ctx.hip; do
  o="$OUT/obj/${f%.*}$SFX.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ]; then /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$o" & fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( objs="$OUT/obj/pfb_spec_$name.o"
    /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs -c pfb_spec.hip -o "$OUT/obj/pfb_spec_$name.o"
    /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs --cuda-device-only -S pfb_spec.hip -o "$OUT/obj/pfb_spec_$name.s" 2>/dev/null
    python3 "$ROOT/tools/check_lds_asm.py" "$OUT/obj/pfb_spec_$name.s" | tail -1
    if [ -n "$AB" ]; then
      /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs -c pfb.hip -o "$OUT/obj/pfb_$name.o"
      /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $defs -c pfb_mfma.hip -o "$OUT/obj/pfb_mfma_$name.o"
      objs="$objs $OUT/obj/pfb_$name.o $OUT/obj/pfb_mfma_$name.o"
    fi
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libsnout_rx_$name.so" $objs \
      "$OUT/obj/btle$SFX.o" "$OUT/obj/zigbee$SFX.o" "$OUT/obj/membench$SFX.o" "$OUT/obj/records$SFX.o" "$OUT/obj/formats$SFX.o" \
      "$OUT/obj/snout_rx$SFX.o" "$OUT/obj/pfb_ctx$SFX.o" && echo "built $name [$defs]" ) &
done
wait
