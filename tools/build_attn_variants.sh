#!/bin/bash
# CPU: measurement builds of the library that differ in attention.hip only (variant macros) -> alpro_amd/lib/variants/libalpro_hip_<v>.so; they travel
# with the snapshot and are timed on one box by tools/attn_bench.py (ALPRO_HIP_LIB selects the build).
#   new        the product
#   head       attention.hip of the last commit (git show HEAD:...)
#   nopipev    -DATTN_NO_PIPE_V        P V with the per-tile fenced operand reads (the form before the rings)
#   pdv2/pdv4  -DATTN_PDV=2 / 4        depth of the V^T operand ring (product: 3)
#   pdk2/pdk6  -DATTN_PDK=2 / 6        depth of the K fragment ring (product: 4)
#   noswap / forcetpl / norot          round-6 first-session variants (ATTN_NO_CLS_SWAP, ATTN_CLS_FORCE_TPL, ATTN_NO_ROT)
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
git show HEAD:alpro_amd/csrc/attention.hip > alpro_amd/csrc/_attention_head.hip
for v in ${@:-new head nopipev pdv2 pdv4 pdk2 pdk6}; do
  case $v in
    new) build new alpro_amd/csrc/attention.hip "" & ;;
    head) build head alpro_amd/csrc/_attention_head.hip "" & ;;
    nopipev) build nopipev alpro_amd/csrc/attention.hip "-DATTN_NO_PIPE_V" & ;;
    pdv2) build pdv2 alpro_amd/csrc/attention.hip "-DATTN_PDV=2" & ;;
    pdv4) build pdv4 alpro_amd/csrc/attention.hip "-DATTN_PDV=4" & ;;
    pdk2) build pdk2 alpro_amd/csrc/attention.hip "-DATTN_PDK=2" & ;;
    pdk6) build pdk6 alpro_amd/csrc/attention.hip "-DATTN_PDK=6" & ;;
    noswap) build noswap alpro_amd/csrc/attention.hip "-DATTN_NO_CLS_SWAP" & ;;
    forcetpl) build forcetpl alpro_amd/csrc/attention.hip "-DATTN_CLS_FORCE_TPL" & ;;
    norot) build norot alpro_amd/csrc/attention.hip "-DATTN_NO_ROT" & ;;
  esac
done
wait
rm -f alpro_amd/csrc/_attention_head.hip
ls -la $V
