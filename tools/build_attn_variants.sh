#!/bin/bash
# CPU: measurement builds of the library that differ in attention.hip only (variant macros) -> alpro_amd/lib/variants/libalpro_hip_<v>.so; they travel
# with the snapshot and are timed on one box (round 6, second session: what do the precise CLS parts still cost the spatial attention forward?)
#   new        the product
#   noswap     -DATTN_NO_CLS_SWAP      the tile with the CLS parts stays with the wave that has two tiles
#   forcetpl   -DATTN_CLS_FORCE_TPL    launches WITHOUT a CLS query run the instantiation that carries the CLS code
#   norot      -DATTN_NO_ROT           no rotation of the tile -> wave walk by workgroup index
set -e
cd "$(dirname "$0")/.."
L=alpro_amd/lib; V=$L/variants; mkdir -p $V
python -m alpro_amd.build > /dev/null 2>&1
OBJS=$(ls $L/obj/*.o | grep -v attention.o)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Werror=inline-asm"
build() {  # name, source, flags
  /opt/rocm/bin/hipcc $FL $3 -Ialpro_amd/csrc -Iinclude -c $2 -o $V/attention_$1.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libalpro_hip_$1.so $OBJS $V/attention_$1.o
  rm $V/attention_$1.o
}
build new alpro_amd/csrc/attention.hip "" &
build noswap alpro_amd/csrc/attention.hip "-DATTN_NO_CLS_SWAP" &
build forcetpl alpro_amd/csrc/attention.hip "-DATTN_CLS_FORCE_TPL" &
build norot alpro_amd/csrc/attention.hip "-DATTN_NO_ROT" &
wait
ls -la $V
