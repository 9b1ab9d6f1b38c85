#!/bin/bash
# round 6, call 6: full GPU suite on the round-6 tree + bench lines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c6
mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_step.json 2> $O/bench_step.err
python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_vfwd.json 2> $O/bench_vfwd.err
python - <<'PY'
import json
for f in ("bench_step","bench_vfwd"):
    try:
        d=json.loads([l for l in open("gpurun_out/r6c6/%s.json"%f) if l.startswith("{")][0])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("divst_subblock",{}).get("ms"), d["roofline"].get("divst_subblock",{}).get("frac"), d["kernel_ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
