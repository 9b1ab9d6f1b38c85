#!/bin/bash
# round 6, call 8: two-stream half-batch forward (ALPRO_SPLIT_STREAMS) A/B on one box + parity of the q-third CLS change
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c8
mkdir -p $O
cd $R
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0"; do
  set -- $cfg
  ALPRO_SPLIT_STREAMS=$1 ALPRO_SPLIT_LOCKSTEP=$2 python bench.py --workload visual_fwd --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-divst > $O/vfwd_split$1_lock$2.json 2>> $O/vfwd.err
  python - "$1" "$2" <<'PY'
import json, sys
s, l = sys.argv[1:3]
try:
    d = json.loads([x for x in open("gpurun_out/r6c8/vfwd_split%s_lock%s.json" % (s, l)) if x.startswith("{")][0])
    print("split", s, "lockstep", l, "ms_per_step", d["ms_per_step"], "clips/s", d["value"])
except Exception as e:
    print("split", s, l, "failed", e)
PY
done
( time ALPRO_SPLIT_STREAMS=1 timeout 1200 python -m pytest tests/test_model_parity.py -m gpu -x -q ) > $O/pytest_parity_split1.log 2>&1
tail -5 $O/pytest_parity_split1.log
