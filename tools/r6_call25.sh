#!/bin/bash
# round 6, call 25: device idle time inside the step (rocprofv3 kernel trace of 7 steps -> tools/gap_report.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c25
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-divst > $O/trace.log 2>&1
F=$(find $O/trace -name '*kernel_trace.csv' | head -1)
head -2 $F > $O/trace_head.txt
python $R/tools/gap_report.py $F > $O/gaps.txt 2>&1
cat $O/gaps.txt
rm -rf $O/trace
