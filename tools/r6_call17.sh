#!/bin/bash
# round 6, call 17: attention forward with operand rings (K fragments / V^T chunks fetched ahead of their MFMA): tests on the product, then the
# variants of tools/build_attn_variants.sh on this box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c17
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -x -q -k "attn or attention" ) > $O/pytest_attn.log 2>&1
tail -3 $O/pytest_attn.log
for rep in 1 2; do
for v in head new nopipev pdv2 pdv4 pdk2 pdk6; do
  echo "== $v" >> $O/variants.txt
  ALPRO_BENCH_DTYPE=fp16 ALPRO_HIP_LIB=$R/alpro_amd/lib/variants/libalpro_hip_$v.so timeout 300 python tools/attn_bench.py fwd 2>>$O/err.log | grep -v "head-major\|dtype\|temporal" >> $O/variants.txt
done
done
cat $O/variants.txt
for v in head new; do
  echo "== $v" >> $O/cls.txt
  ALPRO_HIP_LIB=$R/alpro_amd/lib/variants/libalpro_hip_$v.so timeout 300 python tools/attn_cls_bench.py 2>>$O/err.log >> $O/cls.txt
done
cat $O/cls.txt
