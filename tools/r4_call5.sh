#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "8phase or gemm_bias_act or persistent_partial" > $O/t_ops.txt 2>&1; tail -4 $O/t_ops.txt
for kind in 0 1; do
  ALPRO_GEMM_KIND=$kind timeout 600 python -m pytest tests/test_model_parity.py -m gpu -q -s -k "full_size_pretrain" > $O/t_proxy_kind$kind.txt 2>&1; echo "gemm_kind $kind"; grep -E "B=64 proxy|passed|failed" $O/t_proxy_kind$kind.txt
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_default.json 2> $O/bench_default.err; cut -c1-220 $O/bench_default.json; python -c "import json;d=json.load(open('$O/bench_default.json'));print(d['roofline']['divst_subblock']['ms'], d['roofline']['divst_subblock']['encoder_forward_ms']);print({k:v for k,v in d['roofline'].items() if k in ('achieved','frac')}, d['kernel_ms_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --cls-precise 0 > $O/bench_fp16_plain.json 2> $O/bench_fp16_plain.err; cut -c1-200 $O/bench_fp16_plain.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_visual_cls.json 2>/dev/null; cut -c1-200 $O/bench_visual_cls.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise 0 > $O/bench_visual_plain.json 2>/dev/null; cut -c1-200 $O/bench_visual_plain.json
