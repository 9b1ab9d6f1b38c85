#!/bin/bash
# round 6, call 15: three K-tiles of global loads in flight in the 128 x 128 kernel: GEMM tests, BERT-shape bench and default bench, base vs exp on one box;
# then the tail hand-over at every K (gemm_tail 2) on the encoder forward
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c15
mkdir -p $O
cd $R
L=alpro_amd/lib
cp $L/libalpro_hip_exp.so $L/libalpro_hip.so
( time timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" ) > $O/pytest_gemm.log 2>&1
tail -3 $O/pytest_gemm.log
for v in base exp; do
  cp $L/libalpro_hip_$v.so $L/libalpro_hip.so
  echo "== $v" >> $O/bert_bench.txt
  python tools/gemm_bert_bench.py 2>&1 | grep -v amdgpu | sed 's/| tile256.*//' >> $O/bert_bench.txt
done
cat $O/bert_bench.txt
bash tools/ab_lib.sh 2>&1 | tee $O/ab_step.txt
for t in 1 2 1 2; do
  ALPRO_GEMM_TAIL=$t python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/vfwd_tail$t.json 2>> $O/err.log
  python - $t <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads([x for x in open("gpurun_out/r6c15/vfwd_tail%s.json" % t) if x.startswith("{")][0])
dv = d["roofline"]["divst_subblock"]
print("tail", t, "vfwd", d["ms_per_step"], "divst", dv["ms"], dv["frac"], dv["per_block_us"])
PY
done
