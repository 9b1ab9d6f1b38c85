#!/bin/bash
# Measurement run of the model-level parity tests: every error next to its asserted limit, nothing asserted (tests/test_model_parity.py
# close() under ALPRO_PARITY_REPORT=1).  The table in DESIGN.md section 2 and the tolerances in the tests come from this output.
#   bash tools/parity_report.sh [out.txt]
out=${1:-gpurun_out/parity_report.txt}
mkdir -p "$(dirname "$out")"
ALPRO_PARITY_REPORT=1 python -m pytest tests/test_model_parity.py -m gpu -q -s 2>&1 | grep -E "parity-report|passed|failed|Error" > "$out"
tail -3 "$out"
