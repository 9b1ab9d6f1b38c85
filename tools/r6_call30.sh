#!/bin/bash
# round 6, call 30: which stream ran on which hardware queue, and when, with 4 and with 8 hardware queues (tools/stream_report.py on a kernel trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c30
mkdir -p $O
cd /tmp
for q in 4 8; do
export GPU_MAX_HW_QUEUES=$q
rocprofv3 --kernel-trace -d $O/trace$q -o t --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-divst > $O/trace$q.log 2>&1
F=$(find $O/trace$q -name '*kernel_trace.csv' | head -1)
echo "== GPU_MAX_HW_QUEUES=$q" >> $O/streams.txt
python $R/tools/stream_report.py $F >> $O/streams.txt 2>&1
rm -rf $O/trace$q
done
cat $O/streams.txt
