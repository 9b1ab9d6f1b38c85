#!/bin/bash
# round 6, call 10: prompter pass on its own stream (ALPRO_PROMPTER_STREAM) x split streams, A/B on the training step; bitwise schedule test; determinism / dist tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c10
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_model_parity.py -m gpu -x -q -k "schedules_are_bitwise" ) > $O/pytest_sched.log 2>&1
tail -4 $O/pytest_sched.log
for cfg in "0 0" "1 auto" "0 auto" "1 0" "1 auto" "0 0"; do
  set -- $cfg
  ALPRO_PROMPTER_STREAM=$1 ALPRO_SPLIT_STREAMS=$2 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_ps$1_split$2.json 2>> $O/step.err
  python - "$1" "$2" <<'PY'
import json, sys
a, s = sys.argv[1:3]
try:
    d = json.loads([x for x in open("gpurun_out/r6c10/step_ps%s_split%s.json" % (a, s)) if x.startswith("{")][0])
    print("prompter_stream", a, "split", s, "ms_per_step", d["ms_per_step"], "pairs/s", d["value"])
except Exception as e:
    print("prompter_stream", a, s, "failed", e)
PY
done
( time timeout 1200 python -m pytest tests/test_dist_gpu.py tests/test_amp_gpu.py -m gpu -x -q ) > $O/pytest_dist.log 2>&1
tail -4 $O/pytest_dist.log
