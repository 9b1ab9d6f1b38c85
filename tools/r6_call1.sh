#!/bin/bash
# round 6, call 1: the GPU suite on the phase-A tree + baseline bench lines + attention micro-benchmark (one box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c1
mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_step.json 2> $O/bench_step.err
python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_vfwd.json 2> $O/bench_vfwd.err
ALPRO_BENCH_DTYPE=fp16 python tools/attn_bench.py all > $O/attn_bench.txt 2>&1
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in ("bench_step","bench_vfwd"):
    d=json.loads([l for l in open("gpurun_out/r6c1/%s.json"%f) if l.startswith("{")][0])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("divst_subblock",{}).get("ms"), d["roofline"].get("divst_subblock",{}).get("frac"))
    if "parity" in d: print({k:v for k,v in d["parity"].items() if k in ("meets_bar","meets_bar_at_full_size","full_size_forward","vtc_logits_max_abs_err")})
    print(d.get("exchange"))
PY
cat $O/summary.txt; cat $O/attn_bench.txt
