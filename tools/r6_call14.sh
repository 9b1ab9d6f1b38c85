#!/bin/bash
# round 6, call 14: layernorm_bwd with the dx row fetched up front: kernel bench + same-box A/B of the two libraries under the default bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c14
mkdir -p $O
cd $R
L=alpro_amd/lib
for v in base exp base exp; do
  cp $L/libalpro_hip_$v.so $L/libalpro_hip.so
  echo "== $v" >> $O/ln_bwd_bench.txt
  python tools/ln_bwd_bench.py 64 >> $O/ln_bwd_bench.txt 2>&1
done
cat $O/ln_bwd_bench.txt
bash tools/ab_lib.sh 2>&1 | tee $O/ab_step.txt
