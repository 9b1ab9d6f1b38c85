#!/bin/bash
# End-of-round validation on the GPU box: the full -m gpu suite, smoke(), the default bench line, the parity report, the vendor-GEMM kernel names.
#   bash tools/final_check.sh <tag>
TAG=${1:-final}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/tests_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/parity_report.sh $O/parity_report.txt > /dev/null 2>&1
R=$PWD
for wl in pretrain_step visual_fwd; do
  if [ $wl = pretrain_step ]; then ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-divst"; B=B64; else ARGS="--workload visual_fwd --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst"; B=B32; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/trace_$wl -o t --output-format csv -- python $R/bench.py $ARGS > $R/$O/trace_$wl.log 2>&1)
  cp $(find $O/trace_$wl -name '*kernel_stats.csv' | head -1) $O/${wl}_${B}_kernel_stats.csv 2>/dev/null
  rm -rf $O/trace_$wl
done
ALPRO_BENCH_SHAPES=1 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_visual_fwd_B32.json 2> $O/gemm_shapes_visual_fwd_B32.txt
ALPRO_BENCH_SHAPES=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-divst > /dev/null 2> $O/gemm_shapes_pretrain_step_B64.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/vendor_trace -o v --output-format csv -- python $R/tools/matmul_probe.py > $R/$O/vendor_probe.txt 2>&1)
cut -d, -f1-4 $(find $O/vendor_trace -name '*kernel_stats.csv' | head -1) | head -12 > $O/vendor_kernels.txt 2>/dev/null
rm -rf $O/vendor_trace
tail -n 3 $O/tests_gpu.txt; tail -n 4 $O/smoke.txt; cut -c1-400 $O/bench_default.json; tail -n 2 $O/parity_report.txt
