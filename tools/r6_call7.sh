#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c7
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" ) > $O/pytest_gemm.log 2>&1
tail -6 $O/pytest_gemm.log
timeout 300 python tools/gemm_tail_ab.py > $O/gemm_tail_ab.txt 2>&1
cat $O/gemm_tail_ab.txt
for t in 1 0; do
  ALPRO_GEMM_TAIL=$t python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_vfwd_tail$t.json 2> $O/bench_vfwd_tail$t.err
done
python - <<'PY'
import json
for t in (1, 0):
    try:
        d=json.loads([l for l in open("gpurun_out/r6c7/bench_vfwd_tail%d.json" % t) if l.startswith("{")][0])
        print("tail", t, "vfwd", d["ms_per_step"], d["roofline"]["divst_subblock"]["ms"], d["roofline"]["divst_subblock"]["frac"], d["kernel_ms_per_step"])
    except Exception as e:
        print("tail", t, "failed", e)
PY
