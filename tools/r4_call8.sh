#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "8phase or attn_cls or attn_full or gemm_rows" > $O/t_ops.txt 2>&1; tail -4 $O/t_ops.txt
timeout 200 python tools/fusion_diag.py > $O/fusion_diag.txt 2>&1; sed -n 2,6p $O/fusion_diag.txt
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -s -k "north_star or full_size_pretrain or retrieval_vs_reference or released" > $O/t_parity.txt 2>&1; grep -E "vtc-logit parity|B=64 proxy|passed|failed" $O/t_parity.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_default.json 2> $O/bench_default.err; cut -c1-220 $O/bench_default.json; python -c "import json;d=json.load(open('$O/bench_default.json'));print(d['roofline']['divst_subblock']['ms'], d['roofline']['divst_subblock']['encoder_forward_ms']);print({k:v for k,v in d['roofline'].items() if k in ('achieved','frac')}, d['kernel_ms_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --cls-precise 0 > $O/bench_fp16_plain.json 2> $O/bench_fp16_plain.err; cut -c1-200 $O/bench_fp16_plain.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise 0 > $O/bench_visual_plain.json 2>/dev/null; cut -c1-200 $O/bench_visual_plain.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_visual_cls.json 2>/dev/null; cut -c1-200 $O/bench_visual_cls.json
