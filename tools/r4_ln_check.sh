#!/bin/bash
# focused check of the LayerNorm-backward / reduction kernels: micro-benchmark + their tests (seconds on the GPU box)
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 300 python tools/ln_bwd_bench.py > $O/ln_bwd_bench3.txt 2>&1; grep -v "amdgpu.ids" $O/ln_bwd_bench3.txt
timeout 600 python -m pytest tests/test_hip_bwd_ops.py -m gpu -q -p no:cacheprovider -k "reproducibility or layernorm or gather_cast or emit" > $O/t_ln.txt 2>&1; tail -2 $O/t_ln.txt
