#!/bin/bash
# evidence refresh only (no tests): tools/profile_round.sh r4 = traces + PMC passes + the two bench lines
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
bash tools/profile_round.sh r4 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
python -c "
import json
d=json.load(open('$O/bench_pretrain_step_B64.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['dominant_instance']['achieved'], r['divst_subblock']['ms'], r['divst_subblock']['frac'], r['divst_subblock']['ms_end_to_end'], d['parity']['meets_bar'], d['parity']['vtc_logits_max_abs_err'], d['kernel_ms_per_step'])
v=json.load(open('$O/bench_visual_fwd_B32.json')); print(v['ms_per_step'], v['value'], v['roofline']['divst_subblock']['ms'])"
