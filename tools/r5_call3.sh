#!/bin/bash
# round 5, GPU call 3: scheduler v2.1 under theft, the gathered fusion input (kernel test, model fixtures, step A/B), a fresh ATen-tail profile,
# durations of the model-level test file.
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 120 python tools/sched_smoke.py > $O/smoke.txt 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.txt; exit 1; }
stamp "smoke: $(tail -1 $O/smoke.txt)"
timeout 300 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -q -p no:cacheprovider -x -k "scheduler or gather_seq" > $O/t_ops.txt 2>&1; stamp "ops tests: $(grep -E 'passed|failed|error' $O/t_ops.txt | tail -1)"
timeout 300 python tools/sched_contention.py > $O/sched_contention.txt 2>&1; stamp "sched_contention"; cat $O/sched_contention.txt | cut -c1-200
for gf in 1 0; do
  ALPRO_GATHER_FUSION=$gf timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_gf${gf}.json 2> $O/step_gf${gf}.err
  python -c "import json;d=json.load(open('$O/step_gf${gf}.json'));print('pretrain_step gather_fusion=$gf: %.3f ms %.1f pairs/s family frac %.4f dom %.4f'%(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dominant_instance']['frac']))" 2>&1 | tail -1
done
stamp "pretrain_step A/B"
timeout 300 python tools/aten_tail.py > $O/aten_tail.txt 2>&1; stamp "aten tail"; head -45 $O/aten_tail.txt | cut -c1-200
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -p no:cacheprovider --durations=30 > $O/t_model.txt 2>&1; stamp "model tests: $(grep -E 'passed|failed|error' $O/t_model.txt | tail -1)"; grep -A32 "slowest" $O/t_model.txt | cut -c1-160
