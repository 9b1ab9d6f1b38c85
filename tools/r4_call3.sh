#!/bin/bash
# round-4 GPU call 3: fixed tests of the new kernels, fused precise CLS query, cost A/B, CU-contention table
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "8phase or gemm_rows or attn_cls or attn_full or gemm_bias_act or persistent_partial" > $O/t_ops.txt 2>&1; tail -4 $O/t_ops.txt
timeout 600 python -m pytest tests/test_hip_bwd_ops.py -m gpu -q -k "adamw or facade or key_owned" > $O/t_bwd.txt 2>&1; tail -3 $O/t_bwd.txt
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -s -k "north_star or full_size_pretrain or retrieval_vs_reference" > $O/t_parity_new.txt 2>&1; grep -E "vtc-logit parity|B=64 proxy|passed|failed" $O/t_parity_new.txt
timeout 400 python tools/gemm_kind_ab.py 3 > $O/gemm_kind_ab.txt 2>&1; cat $O/gemm_kind_ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json; python -c "import json;d=json.load(open('$O/bench_default.json'));print(d['parity']['vtc_logits_max_abs_err_per_fixture'], d['parity']['meets_bar']);print(d['roofline']['divst_subblock']);print(d['roofline']['dominant_instance']);print({k:v for k,v in d['roofline'].items() if k in ('achieved','frac')}, d['kernel_ms_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --cls-precise 0 > $O/bench_fp16_plain.json 2> $O/bench_fp16_plain.err; cut -c1-200 $O/bench_fp16_plain.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_visual_cls.json 2>/dev/null; cut -c1-200 $O/bench_visual_cls.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise 0 > $O/bench_visual_plain.json 2>/dev/null; cut -c1-200 $O/bench_visual_plain.json
timeout 300 python tools/overlap_contention.py --steps 4 > $O/overlap_cu_contention.txt 2>&1; tail -28 $O/overlap_cu_contention.txt
timeout 600 python -m pytest tests/test_amp_gpu.py tests/test_dist_gpu.py -m gpu -q -x > $O/t_regress.txt 2>&1; tail -3 $O/t_regress.txt
