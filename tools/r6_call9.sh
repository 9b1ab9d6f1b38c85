#!/bin/bash
# round 6, call 9: deferred temporal add (ALPRO_DEFER_TEMPORAL_ADD) A/B with the divST table, bitwise test of both inference schedules
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c9
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_model_parity.py -m gpu -x -q -k "schedules_are_bitwise or cls_stream or fused_temporal" ) > $O/pytest_sched.log 2>&1
tail -5 $O/pytest_sched.log
for cfg in "0 auto" "1 auto" "0 0" "1 0" "1 auto"; do
  set -- $cfg
  ALPRO_DEFER_TEMPORAL_ADD=$1 ALPRO_SPLIT_STREAMS=$2 python bench.py --workload visual_fwd --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/vfwd_defer$1_split$2.json 2>> $O/vfwd.err
  python - "$1" "$2" <<'PY'
import json, sys
d_, s = sys.argv[1:3]
try:
    d = json.loads([x for x in open("gpurun_out/r6c9/vfwd_defer%s_split%s.json" % (d_, s)) if x.startswith("{")][0])
    dv = d["roofline"]["divst_subblock"]
    print("defer", d_, "split", s, "ms_per_step", d["ms_per_step"], "divst ms", dv["ms"], "frac", dv["frac"], "per block", dv["measured_us_per_block"])
    print("   ", dv["per_block_us"])
except Exception as e:
    print("defer", d_, s, "failed", e)
PY
done
tail -3 $O/vfwd.err
