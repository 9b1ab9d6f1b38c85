#!/bin/bash
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4f; mkdir -p $O
timeout 400 python tools/proxy_diag.py > $O/proxy_diag.txt 2>&1; tail -9 $O/proxy_diag.txt
cd /tmp
for wl in visual_fwd pretrain_step; do
  if [ $wl = pretrain_step ]; then ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-divst"; else ARGS="--workload visual_fwd --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst"; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$wl -o t --output-format csv -- python $R/bench.py $ARGS > $O/trace_$wl.log 2>&1
  cp $(find $O/trace_$wl -name '*kernel_stats.csv' | head -1) $O/${wl}_kernel_stats.csv 2>/dev/null
  rm -rf $O/trace_$wl
  echo "== $wl"; head -32 $O/${wl}_kernel_stats.csv | cut -c1-200
done
