#!/bin/bash
# MI355X counterpart of the reference's run_scripts/pt_alpro.sh: the UNCHANGED pretraining driver, one process per GPU over RCCL.
#   REFERENCE=/path/to/ALPRO bash run_scripts/pt_alpro_mi355x.sh [extra driver args]
set -e
REF=${REFERENCE:?set REFERENCE to the salesforce/ALPRO checkout}
NPROC=${NPROC:-8}
cd "$(dirname "$0")/.."
python -m alpro_amd.launch --nproc "$NPROC" --reference "$REF" src/pretrain/run_pretrain_sparse.py \
    --config config_release/pretrain_alpro.json --output_dir "${OUTPUT_DIR:-/tmp/alpro_pretrain_$(date +%Y%m%d%H%M%S)}" "$@"
