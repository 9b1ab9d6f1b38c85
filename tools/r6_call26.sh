#!/bin/bash
# round 6, call 26: does the number of hardware queues the HIP runtime maps streams onto matter for the multi-stream step?  GPU_MAX_HW_QUEUES unset (4) / 8 / 2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c26
mkdir -p $O
cd $R
for i in 1 2; do
for v in default 8 2; do
if [ $v = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$v; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_${v}_$i.json 2>> $O/err.log
python - $v $i <<'PY'
import json, sys
d = json.loads([x for x in open("gpurun_out/r6c26/step_%s_%s.json" % (sys.argv[1], sys.argv[2])) if x.startswith("{")][0])
print("hw_queues", sys.argv[1], "step ms", d["ms_per_step"], d["value"])
PY
done
done
unset GPU_MAX_HW_QUEUES
for v in default 8; do
if [ $v = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$v; fi
python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-divst 2>/dev/null | python -c "import sys,json; d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('visual_fwd hw_queues $v', d['ms_per_step'], d['value'])"
done
