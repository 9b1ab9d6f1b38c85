#!/bin/bash
# round 6, call 33: last check of the final tree: smoke(), the whole GPU suite, the default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c33
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([x for x in open("gpurun_out/r6c33/bench.json") if x.startswith("{")][0])
r = d["roofline"]
print("bench", d["value"], d["ms_per_step"], r["frac"], r["dominant_instance"]["frac"], r["divst_subblock"]["frac"], r["divst_subblock"]["ms"])
PY
