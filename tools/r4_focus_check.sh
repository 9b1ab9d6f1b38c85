#!/bin/bash
# round 4, GPU call 10: ragged tile row inside the 8-phase kernel + LayerNorm-backward grid cap -- focused tests, GEMM A/B table, bench
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -q -p no:cacheprovider -k "8phase or reproducibility or layernorm or gather_cast or full_benchmark_size or gelu" > $O/t_ops.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/t_ops.txt | head -20
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -s -p no:cacheprovider -k "north_star or full_size_pretrain" > $O/t_parity.txt 2>&1; grep -E "vtc-logit parity|B=64 proxy|passed|failed" $O/t_parity.txt
timeout 300 python tools/gemm_kind_ab.py > $O/gemm_kind_ab.txt 2>&1; grep -v amdgpu.ids $O/gemm_kind_ab.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_default.json 2> $O/bench_default.err; cut -c1-220 $O/bench_default.json; python -c "import json;d=json.load(open('$O/bench_default.json'));print(d['roofline']['divst_subblock']['ms'], d['roofline']['divst_subblock']['encoder_forward_ms']);print({k:v for k,v in d['roofline'].items() if k in ('achieved','frac')}, d['kernel_ms_per_step'])"
