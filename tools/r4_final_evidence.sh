#!/bin/bash
# round 4, final GPU call: evidence with the FINAL binary -- kernel traces, PMC passes, bench lines (tools/profile_round.sh r4), the per-mode
# step times of the parity Pareto table, the GPU suite, smoke().  (The model-level files tests/test_model_parity.py, test_amp_gpu.py and
# test_dist_gpu.py take 8 minutes; when they have just run on the same binary -- tools/r4_focus_check.sh -- pass "ops" to skip them.)
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
bash tools/profile_round.sh r4 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
cut -c1-260 $O/bench_pretrain_step_B64.json
for mode in "fp16 0" "bf16 0" "bf16 1" "fp32 0"; do
  set -- $mode
  steps=8; [ $1 = fp32 ] && steps=2
  timeout 600 python bench.py --dtype $1 --cls-precise $2 --steps $steps --warmup 2 --no-cpu-baseline --no-parity --no-divst > $O/bench_mode_$1_cls$2.json 2> $O/bench_mode_$1_cls$2.err
  python -c "import json;d=json.load(open('$O/bench_mode_$1_cls$2.json'));print('$1 cls=$2', d['mode'], d['ms_per_step'], 'ms', d['value'], d['unit'])"
done
if [ "$1" = ops ] || [ "$ALPRO_EVIDENCE_TESTS" = ops ]; then
  timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py tests/test_input_gpu.py tests/test_bench_multirank.py -m gpu -q -p no:cacheprovider > $O/t_ops_final.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/t_ops_final.txt | head
else
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/t_all.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed|vtc-logit parity|B=64 proxy" $O/t_all.txt | head -40
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
