#!/bin/bash
# round 6, call 21: the frozen prompter's pass forked behind the visual forward (ALPRO_PROMPTER_STREAM=1): bitwise check at B = 8, then A/B/A/B of the B = 64 step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c21
mkdir -p $O
cd $R
timeout 900 python tools/wgrad_stream_check.py 8 > $O/bitwise.txt 2>&1
tail -6 $O/bitwise.txt
for i in 1 2; do
for v in 0 1; do
ALPRO_PROMPTER_STREAM=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_${v}_$i.json 2>> $O/err.log
python - $v $i <<'PY'
import json, sys
d = json.loads([x for x in open("gpurun_out/r6c21/step_%s_%s.json" % (sys.argv[1], sys.argv[2])) if x.startswith("{")][0])
print("prompter_stream", sys.argv[1], "step ms", d["ms_per_step"], d["value"], "peak GB", d["peak_mem_gb"], d.get("wgrad_side_stream"), d.get("text_side_stream"), d.get("prompter_side_stream"))
PY
done
done
tail -3 $O/err.log
