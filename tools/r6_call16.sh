#!/bin/bash
# round 6, call 16: AdamW with two chunks in flight + the 16-bit mirror written by the same pass: tests, then the default bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c16
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_hip_bwd_ops.py tests/test_amp_gpu.py tests/test_dist_gpu.py -m gpu -x -q -k "adamw or optim or amp or scal or bitwise or rank or step" ) > $O/pytest_opt.log 2>&1
tail -4 $O/pytest_opt.log
for i in 1 2; do
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step$i.json 2>> $O/err.log
python - $i <<'PY'
import json, sys
d = json.loads([x for x in open("gpurun_out/r6c16/step%s.json" % sys.argv[1]) if x.startswith("{")][0])
print("step ms", d["ms_per_step"], d["value"], d["kernel_ms_per_step"])
PY
done
