#!/bin/bash
# round-4 GPU call 1: the precise-CLS-row mode (kernel test, fixtures, B=64 proxy, cost A/B), the optimizer trajectory, plumbing regressions
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attn_cls" > $O/t_attn_cls.txt 2>&1; tail -3 $O/t_attn_cls.txt
timeout 300 python -m pytest tests/test_hip_bwd_ops.py -m gpu -q -k "adamw or facade" > $O/t_adamw.txt 2>&1; tail -3 $O/t_adamw.txt
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -s -k "north_star or full_size_pretrain" > $O/t_parity_new.txt 2>&1; grep -E "vtc-logit parity|B=64 proxy|passed|failed" $O/t_parity_new.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_fp16_cls.json 2> $O/bench_fp16_cls.err; cut -c1-300 $O/bench_fp16_cls.json; python -c "import json;d=json.load(open('$O/bench_fp16_cls.json'));print(d['parity'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --cls-precise 0 > $O/bench_fp16_plain.json 2> $O/bench_fp16_plain.err; cut -c1-300 $O/bench_fp16_plain.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_visual_cls.json 2>/dev/null; cut -c1-200 $O/bench_visual_cls.json
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise 0 > $O/bench_visual_plain.json 2>/dev/null; cut -c1-200 $O/bench_visual_plain.json
timeout 900 python -m pytest tests/test_model_parity.py tests/test_amp_gpu.py tests/test_dist_gpu.py -m gpu -q -x -k "not north_star and not full_size_pretrain" > $O/t_regress.txt 2>&1; tail -4 $O/t_regress.txt
