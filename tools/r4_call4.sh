#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
timeout 400 python tools/gemm_kind_check.py > $O/gemm_kind_check.txt 2>&1; cut -c1-330 $O/gemm_kind_check.txt
timeout 600 python -m pytest tests/test_hip_bwd_ops.py -m gpu -q -k "gemm_tn_acc" > $O/t_tn.txt 2>&1; tail -4 $O/t_tn.txt
timeout 400 python tools/gemm_tn_kind_ab.py 3 > $O/gemm_tn_kind_ab.txt 2>&1; cat $O/gemm_tn_kind_ab.txt
