#!/bin/bash
# round 6, call 22: the precise-CLS chain of the TRAINING forward on its side stream (ALPRO_CLS_STREAM=1) against the default (infer), with this
# session's side streams on; A/B/A/B of the B = 64 step on this box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c22
mkdir -p $O
cd $R
for i in 1 2; do
for v in infer 1; do
ALPRO_CLS_STREAM=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_${v}_$i.json 2>> $O/err.log
python - $v $i <<'PY'
import json, sys
d = json.loads([x for x in open("gpurun_out/r6c22/step_%s_%s.json" % (sys.argv[1], sys.argv[2])) if x.startswith("{")][0])
print("cls_stream", sys.argv[1], "step ms", d["ms_per_step"], d["value"], "peak GB", d["peak_mem_gb"])
PY
done
done
tail -3 $O/err.log
