#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_model_parity.py tests/test_amp_gpu.py tests/test_dist_gpu.py -m gpu -q -p no:cacheprovider -x > $O/t_model.txt 2>&1; tail -3 $O/t_model.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/bench_default2.json 2> $O/bench_default2.err; python -c "import json;d=json.load(open('$O/bench_default2.json'));print(d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])"
