#!/bin/bash
# round 6, call 27: GPU_MAX_HW_QUEUES 3 / 4 / 5 / 6 under the multi-stream step (call 26: 8 is 12 % slower than the default 4, 2 is 0.6 % slower)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c27
mkdir -p $O
cd $R
for i in 1 2; do
for v in 4 3 5 6; do
export GPU_MAX_HW_QUEUES=$v
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst 2>> $O/err.log | python -c "import sys,json; d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('hw_queues $v step ms', d['ms_per_step'], d['value'])"
done
done
