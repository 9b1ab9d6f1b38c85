#!/bin/bash
# round 5, GPU call 5: early tickets, re-check of the failed Block test, step + ATen tail with the fusion-output node and the ordered scatter.
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 120 python tools/sched_smoke.py > $O/smoke.txt 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.txt; exit 1; }
stamp "smoke: $(tail -1 $O/smoke.txt)"
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -q -p no:cacheprovider -k "merged_temporal or scheduler or 8phase or gelu or deterministic or reductions" > $O/t_ops.txt 2>&1; stamp "ops tests: $(grep -E 'passed|failed|error' $O/t_ops.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/t_ops.txt | head
timeout 300 python tools/sched_contention.py > $O/sched_contention.txt 2>&1; stamp "sched_contention"; cat $O/sched_contention.txt | cut -c1-200
for i in 1 2; do
ALPRO_BENCH_SHAPES=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step$i.json 2> $O/step_shapes$i.txt
python -c "import json;d=json.load(open('$O/step$i.json'));print('pretrain_step: %.3f ms %.1f pairs/s family frac %.4f dom %.4f'%(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dominant_instance']['frac']))" 2>&1 | tail -1
done
ALPRO_GEMM_SCHED=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_sc0.json 2> /dev/null
python -c "import json;d=json.load(open('$O/step_sc0.json'));print('pretrain_step sched=0: %.3f ms %.1f pairs/s'%(d['ms_per_step'],d['value']))" 2>&1 | tail -1
stamp "pretrain_step"
timeout 300 python tools/aten_tail.py > $O/aten_tail.txt 2>&1; stamp "aten tail"; grep -A22 "non-alpro device time" $O/aten_tail.txt | cut -c1-170
