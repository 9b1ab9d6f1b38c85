#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
timeout 300 python tools/fusion_diag.py > $O/fusion_diag.txt 2>&1; tail -14 $O/fusion_diag.txt
DIAG_B=3 timeout 300 python tools/fusion_diag.py > $O/fusion_diag_B3.txt 2>&1; tail -14 $O/fusion_diag_B3.txt | head -6
