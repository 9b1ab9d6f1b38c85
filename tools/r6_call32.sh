#!/bin/bash
# round 6, call 32: the last fusion layer's row-wise tail on the rows the heads read only (ALPRO_FUSION_TAIL_ROWS=1) against every row (0), A/B/A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c32
mkdir -p $O
cd $R
for i in 1 2; do
for v in 0 1; do
ALPRO_FUSION_TAIL_ROWS=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst 2>> $O/err.log | python -c "import sys,json; d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('fusion_tail_rows $v step ms', d['ms_per_step'], d['value'], 'peak GB', d['peak_mem_gb'])"
done
done
