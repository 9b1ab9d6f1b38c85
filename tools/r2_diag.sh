#!/bin/bash
# Round-2 diagnostic pass: per-shape GEMM timings, PMC counters on the dominant GEMM shapes, visual_fwd baseline.
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 -L > $O/counters.txt 2>&1
python tools/gemm_bench.py 256 > $O/gemm_bench.txt 2>&1
python bench.py --workload visual_fwd --steps 10 --warmup 3 --no-cpu-baseline > $O/visual_fwd.json 2> $O/visual_fwd.err
for shape in "50208 2304 768" "50208 768 768" "50208 768 3072"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/pmc1_$tag -o pmc --output-format csv -- python tools/gemm_one.py $shape 256 4 > $O/pmc1_$tag.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/pmc2_$tag -o pmc --output-format csv -- python tools/gemm_one.py $shape 256 4 > $O/pmc2_$tag.log 2>&1
done
find $O -name '*counter_collection.csv' | while read f; do echo "== $f"; python tools/pmc_raw.py $f gemm; done > $O/pmc_summary.txt 2>&1
# keep the merged output small
find $O -name '*.csv' -size +2M -delete
ls -R $O | head -50
