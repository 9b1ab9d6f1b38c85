#!/bin/bash
# Issue / wait / LDS counters of the spatial attention backward, two-phase kernel (ALPRO_ATTN_BWD=0) next to the key-owned one (=1).
# Run through gpurun: bash tools/attn_counters.sh [tag]; separate --pmc passes, no traces.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attnc${1:+_$1}
mkdir -p $O
cd /tmp
run() {  # tag, counters..., then -- cmd
  tag=$1; shift
  ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rocprofv3 --pmc "${ctr[@]}" -d $O/$tag -o c --output-format csv -- "$@" > $O/$tag.log 2>&1
  python $R/tools/pmc_raw.py $(find $O/$tag -name '*counter_collection.csv' | head -1) attn_bwd > $O/$tag.txt 2>&1
}
for k in ${KINDS:-0 1}; do
  export ALPRO_ATTN_BWD=$k
  CMD="python $R/tools/attn_bwd_one.py"
  run k${k}_wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
  run k${k}_inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $CMD
done
find $O -name '*.db' -delete; find $O -name '*.csv' -size +1000k -delete
for f in $O/*.txt; do echo "== $f"; cat $f; done
