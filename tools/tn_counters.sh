#!/bin/bash
# LDS / issue counters of the weight-gradient kernel next to the NT kernel on a same-size problem (run through gpurun).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tnc
mkdir -p $O
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/avail_sq.txt
run() {  # tag, counters..., then -- cmd
  tag=$1; shift
  ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rocprofv3 --pmc "${ctr[@]}" -d $O/$tag -o c --output-format csv -- "$@" > $O/$tag.log 2>&1
  python $R/tools/pmc_raw.py $(find $O/$tag -name '*counter_collection.csv' | head -1) gemm_ > $O/$tag.txt 2>&1
}
for k in tn nt; do
  if [ $k = tn ]; then CMD="python $R/tools/gemm_tn_bench.py 2"; else CMD="python $R/tools/gemm_one.py 100416 3072 768"; fi
  run ${k}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS -- $CMD
  run ${k}_wait SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES -- $CMD
  run ${k}_inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -- $CMD
done
find $O -name '*.db' -delete; find $O -name '*.csv' -size +1000k -delete
cat $O/*_lds.txt $O/*_wait.txt $O/*_inst.txt
