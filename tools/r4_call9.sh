#!/bin/bash
# round 4, GPU call 9: the whole GPU suite on the binary with the deterministic reductions + staging fences, then the default bench
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/t_all.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/t_all.txt | head -40
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_default.json 2> $O/bench_default.err; cut -c1-220 $O/bench_default.json; python -c "import json;d=json.load(open('$O/bench_default.json'));print(d['roofline']['divst_subblock']['ms'], d['roofline']['divst_subblock']['encoder_forward_ms']);print({k:v for k,v in d['roofline'].items() if k in ('achieved','frac')}, d['kernel_ms_per_step'])"
ALPRO_DETERMINISTIC=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/bench_atomics.json 2> $O/bench_atomics.err; cut -c1-200 $O/bench_atomics.json
