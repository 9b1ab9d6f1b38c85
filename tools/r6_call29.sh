#!/bin/bash
# round 6, call 29: the text backward's weight gradients kept on the text side stream (no side stream of a side stream): 4 / 8 / 6 hardware queues,
# against the nested form (ALPRO_WGRAD_STREAM_NESTED=1, the form of calls 20-28)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c29
mkdir -p $O
cd $R
run() {  # label, env assignments...
  lbl=$1; shift
  env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-divst 2>> $O/err.log | python -c "import sys,json; d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('$lbl step ms', d['ms_per_step'], d['value'])"
}
for i in 1 2; do
run "q4 flat  " GPU_MAX_HW_QUEUES=4
run "q4 nested" GPU_MAX_HW_QUEUES=4 ALPRO_WGRAD_STREAM_NESTED=1
run "q8 flat  " GPU_MAX_HW_QUEUES=8
run "q8 nested" GPU_MAX_HW_QUEUES=8 ALPRO_WGRAD_STREAM_NESTED=1
run "q6 flat  " GPU_MAX_HW_QUEUES=6
run "q5 flat  " GPU_MAX_HW_QUEUES=5
done
