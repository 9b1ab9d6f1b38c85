#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in base nops vol k32; do
  echo "== $v"
  ALPRO_HIP_LIB=$R/alpro_amd/lib/variants/libalpro_hip_$v.so python tools/tattn_diag.py 2>&1 | grep "rep 0\|training" | cut -c1-200
done > gpurun_out/tattn_variants.txt 2>&1
cat gpurun_out/tattn_variants.txt
