#!/bin/bash
# round 6, call 12: the CLS tile on the wave with a round less (attention forward): tests, divST table, training step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c12
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "attn" ) > $O/pytest_attn.log 2>&1
tail -4 $O/pytest_attn.log
python bench.py --workload visual_fwd --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/vfwd.json 2>> $O/err.log
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step.json 2>> $O/err.log
python - <<'PY'
import json
d = json.loads([x for x in open("gpurun_out/r6c12/vfwd.json") if x.startswith("{")][0])
dv = d["roofline"]["divst_subblock"]
print("vfwd ms_per_step", d["ms_per_step"], "divst ms", dv["ms"], "frac", dv["frac"], "per block", dv["measured_us_per_block"])
print("   ", dv["per_block_us"])
d = json.loads([x for x in open("gpurun_out/r6c12/step.json") if x.startswith("{")][0])
print("step ms", d["ms_per_step"], d["value"], d["kernel_ms_per_step"])
PY
( time timeout 900 python -m pytest tests/test_model_parity.py -m gpu -x -q -k "cls or precise or retrieval or pretrain" ) > $O/pytest_parity.log 2>&1
tail -4 $O/pytest_parity.log
