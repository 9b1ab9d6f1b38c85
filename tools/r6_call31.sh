#!/bin/bash
# round 6, call 31: HIP priority of the side streams (text, weight gradients, prompter): default / high / low, A/B/C twice
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c31
mkdir -p $O
cd $R
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())"
for i in 1 2; do
for v in default high low; do
ALPRO_SIDE_PRIORITY=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-divst 2>> $O/err.log | python -c "import sys,json; d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('side priority $v step ms', d['ms_per_step'], d['value'])"
done
done
