#!/bin/bash
# round 6, call 28: which pair of streams must NOT run beside each other?  With GPU_MAX_HW_QUEUES=8 (every stream its own hardware queue: the step is
# 12 % slower than with the default 4) each side-stream feature is switched off in turn.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c28
mkdir -p $O
cd $R
run() {  # label, env assignments...
  lbl=$1; shift
  env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-divst 2>> $O/err.log | python -c "import sys,json; d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('$lbl step ms', d['ms_per_step'], d['value'])"
}
run "q4 all-on      " GPU_MAX_HW_QUEUES=4
run "q8 all-on      " GPU_MAX_HW_QUEUES=8
run "q8 wgrad-off   " GPU_MAX_HW_QUEUES=8 ALPRO_WGRAD_STREAM=0
run "q8 text-off    " GPU_MAX_HW_QUEUES=8 ALPRO_TEXT_STREAM=0
run "q8 prompter-off" GPU_MAX_HW_QUEUES=8 ALPRO_PROMPTER_STREAM=0
run "q8 split-off   " GPU_MAX_HW_QUEUES=8 ALPRO_SPLIT_STREAMS=0
run "q8 cls-off     " GPU_MAX_HW_QUEUES=8 ALPRO_CLS_STREAM=0
run "q8 all-off     " GPU_MAX_HW_QUEUES=8 ALPRO_WGRAD_STREAM=0 ALPRO_TEXT_STREAM=0 ALPRO_PROMPTER_STREAM=0 ALPRO_SPLIT_STREAMS=0 ALPRO_CLS_STREAM=0
run "q4 all-off     " GPU_MAX_HW_QUEUES=4 ALPRO_WGRAD_STREAM=0 ALPRO_TEXT_STREAM=0 ALPRO_PROMPTER_STREAM=0 ALPRO_SPLIT_STREAMS=0 ALPRO_CLS_STREAM=0
