#!/bin/bash
# round 6, call 2: new spatial attention forward (CLS parts through the MFMA path, packed softmax, K-then-V waits)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c2
mkdir -p $O
cd $R
( time python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -x -q -k "attn or attention" ) > $O/pytest_attn.log 2>&1
tail -5 $O/pytest_attn.log
ALPRO_BENCH_DTYPE=fp16 python tools/attn_bench.py fwd > $O/attn_bench.txt 2>&1
cat $O/attn_bench.txt
( time python -m pytest tests/test_model_parity.py -m gpu -x -q ) > $O/pytest_model.log 2>&1
tail -5 $O/pytest_model.log
python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_vfwd.json 2> $O/bench_vfwd.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6c2/bench_vfwd.json") if l.startswith("{")][0])
print("vfwd", d["ms_per_step"], d["roofline"]["divst_subblock"]["ms"], d["roofline"]["divst_subblock"]["frac"])
PY
