#!/bin/bash
# One round's profile evidence (run on the GPU box through gpurun):  bash tools/profile_round.sh <tag, e.g. r2>
#   - (last, so that they can cite this round's PMC bytes) default bench line (pretrain_step B=64) and visual_fwd B=32 bench line, with the per-shape GEMM table
#   - rocprofv3 --kernel-trace --stats of both commands -> per-kernel stats CSV
#   - separate --pmc passes (never combined with traces): FETCH_SIZE, WRITE_SIZE (HBM traffic) and the MFMA-utilisation counters
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
TAG=${1:-r3}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cd /tmp
for wl in pretrain_step visual_fwd; do
  # clean, step-delimited traces: no divST pass, no parity model, no CPU baseline inside the traced process -> every per-step
  # kernel appears (warmup + steps + 1 per-kernel timing pass) = 8 (pretrain_step) / 14 (visual_fwd) times its per-step launch count
  if [ $wl = pretrain_step ]; then ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-divst"; B=B64; else ARGS="--workload visual_fwd --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst"; B=B32; fi
  rocprofv3 --kernel-trace --stats -d $O/trace_$wl -o t --output-format csv -- python $R/bench.py $ARGS > $O/trace_$wl.log 2>&1
  cp $(find $O/trace_$wl -name '*kernel_stats.csv' | head -1) $O/${wl}_${B}_kernel_stats.csv 2>/dev/null
  if [ $wl = visual_fwd ]; then continue; fi   # counters for the training step only (GPU-minute budget)
  for pm in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pm -d $O/pmc_${wl}_$pm -o $pm --output-format csv -- python $R/bench.py $ARGS > $O/pmc_${wl}_$pm.log 2>&1
    mkdir -p $O/pmc_$wl && cp $(find $O/pmc_${wl}_$pm -name '*counter_collection.csv' | head -1) $O/pmc_$wl/${pm}_counter_collection.csv 2>/dev/null
  done
  python $R/tools/pmc_summary.py $O/pmc_$wl $O/${wl}_${B}_pmc_traffic.json > $O/${wl}_${B}_pmc_traffic.txt 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_${wl}_mfma -o mfma --output-format csv -- python $R/bench.py $ARGS > $O/pmc_${wl}_mfma.log 2>&1
  python $R/tools/mfma_summary.py $(find $O/pmc_${wl}_mfma -name '*counter_collection.csv' | head -1) > $O/${wl}_${B}_mfma_util.txt 2>&1
done
# the bench lines LAST: bench.py reads the PMC bytes per GEMM launch from the latest profiles/r*_pretrain_step_B64_pmc_traffic.json, i.e. from
# the passes above (copied into profiles/ here on the box; the same file comes home through gpurun_out/ and is committed under that name)
cd $R
cp $O/pretrain_step_B64_pmc_traffic.json profiles/${TAG}_pretrain_step_B64_pmc_traffic.json 2>/dev/null
ALPRO_BENCH_SHAPES=1 python bench.py --steps 10 --warmup 3 > $O/bench_pretrain_step_B64.json 2> $O/gemm_shapes_pretrain_step_B64.txt
ALPRO_BENCH_SHAPES=1 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_visual_fwd_B32.json 2> $O/gemm_shapes_visual_fwd_B32.txt
find $O -name '*.csv' -size +1500k -delete
find $O -name '*.db' -delete
du -sh $O; ls $O
