#!/bin/bash
# round-4 GPU call 2: 8-phase GEMM (correctness + A/B), skinny fp32 GEMM, precise-CLS mode again (fixed B=1 aliasing; cost with the new Linears)
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "8phase or gemm_rows" > $O/t_gemm_new.txt 2>&1; tail -5 $O/t_gemm_new.txt
timeout 600 python tools/gemm_kind_ab.py 5 > $O/gemm_kind_ab.txt 2>&1; cat $O/gemm_kind_ab.txt
timeout 300 python -m pytest tests/test_hip_bwd_ops.py -m gpu -q -k "adamw or facade" > $O/t_adamw.txt 2>&1; tail -3 $O/t_adamw.txt
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -s -k "north_star or full_size_pretrain or retrieval_vs_reference or forward_cls" > $O/t_parity_new.txt 2>&1; grep -E "vtc-logit parity|B=64 proxy|passed|failed" $O/t_parity_new.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/bench_fp16_cls.json 2> $O/bench_fp16_cls.err; cut -c1-300 $O/bench_fp16_cls.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --cls-precise 0 > $O/bench_fp16_plain.json 2> $O/bench_fp16_plain.err; cut -c1-300 $O/bench_fp16_plain.json
ALPRO_GEMM_KIND=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --cls-precise 0 > $O/bench_fp16_plain_kind1.json 2> $O/bench_fp16_plain_kind1.err; cut -c1-300 $O/bench_fp16_plain_kind1.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_visual_cls.json 2>/dev/null; cut -c1-200 $O/bench_visual_cls.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise 0 > $O/bench_visual_plain.json 2>/dev/null; cut -c1-200 $O/bench_visual_plain.json
ALPRO_GEMM_KIND=1 timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise 0 > $O/bench_visual_plain_kind1.json 2>/dev/null; cut -c1-200 $O/bench_visual_plain_kind1.json
