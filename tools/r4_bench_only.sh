#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
cp $O/pretrain_step_B64_pmc_traffic.json profiles/r4_pretrain_step_B64_pmc_traffic.json 2>/dev/null
ALPRO_BENCH_SHAPES=1 python bench.py --steps 10 --warmup 3 > $O/bench_pretrain_step_B64.json 2> $O/gemm_shapes_pretrain_step_B64.txt
ALPRO_BENCH_SHAPES=1 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_visual_fwd_B32.json 2> $O/gemm_shapes_visual_fwd_B32.txt
python -c "
import json
d=json.load(open('$O/bench_pretrain_step_B64.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['dominant_instance']['achieved'], r['divst_subblock']['ms'], r['divst_subblock']['frac'], r['divst_subblock']['ms_end_to_end'], d['parity']['meets_bar'], d['parity']['vtc_logits_max_abs_err'])
v=json.load(open('$O/bench_visual_fwd_B32.json')); print(v['ms_per_step'], v['value'], v['roofline']['divst_subblock']['ms'])"
