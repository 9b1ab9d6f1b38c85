#!/bin/bash
# round 6, call 13: gemm_rows LN statistics from registers; whole GPU suite; both bench lines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c13
mkdir -p $O
cd $R
python bench.py --workload visual_fwd --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/vfwd.json 2>> $O/err.log
python - <<'PY'
import json
d = json.loads([x for x in open("gpurun_out/r6c13/vfwd.json") if x.startswith("{")][0])
dv = d["roofline"]["divst_subblock"]
print("vfwd ms_per_step", d["ms_per_step"], "divst ms", dv["ms"], "frac", dv["frac"], "per block", dv["measured_us_per_block"])
print("   ", dv["per_block_us"])
PY
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/step.json 2>> $O/err.log
python - <<'PY'
import json
d = json.loads([x for x in open("gpurun_out/r6c13/step.json") if x.startswith("{")][0])
print("step ms", d["ms_per_step"], d["value"], d["kernel_ms_per_step"])
print(d["roofline"].get("divst_subblock"))
print(d.get("parity"))
PY
