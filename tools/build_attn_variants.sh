#!/bin/bash
# CPU: measurement builds of the library that differ in attention.hip only (variant macros ATTN_NO_ROT / ATTN_NO_KV_SPLIT / ATTN_NO_PK, and the
# round-5 kernel) -> alpro_amd/lib/variants/libalpro_hip_<v>.so; they travel with the snapshot, tools/r6_call3.sh times them on one box.
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
git show 0f6f40f:alpro_amd/csrc/attention.hip > /tmp/attention_r5.hip
build new alpro_amd/csrc/attention.hip "" &
build norot alpro_amd/csrc/attention.hip "-DATTN_NO_ROT" &
build nokv alpro_amd/csrc/attention.hip "-DATTN_NO_KV_SPLIT" &
build nopk alpro_amd/csrc/attention.hip "-DATTN_NO_PK" &
wait
build nokvrot alpro_amd/csrc/attention.hip "-DATTN_NO_KV_SPLIT -DATTN_NO_ROT" &
build r5 /tmp/attention_r5.hip "-Wno-error=inline-asm" &
wait
ls -la $V
