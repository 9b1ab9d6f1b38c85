#!/bin/bash
# round 5, GPU call 4: the epilogue rework (stores without per-pass waits, saved-factor run-ahead, one-exponential GELU), fair-share tickets, ordered
# scatter, fusion outputs -- op tests, model tests, per-shape GEMM table, step, contention.
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 120 python tools/sched_smoke.py > $O/smoke.txt 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.txt; exit 1; }
stamp "smoke: $(tail -1 $O/smoke.txt)"
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -q -p no:cacheprovider --durations=25 > $O/t_ops.txt 2>&1; stamp "ops tests: $(grep -E 'passed|failed|error' $O/t_ops.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/t_ops.txt | head; grep -A27 "slowest" $O/t_ops.txt | cut -c1-150
timeout 300 python tools/sched_contention.py > $O/sched_contention.txt 2>&1; stamp "sched_contention"; cat $O/sched_contention.txt | cut -c1-200
ALPRO_BENCH_SHAPES=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/step.json 2> $O/step_shapes.txt
python -c "import json;d=json.load(open('$O/step.json'));print('pretrain_step: %.3f ms %.1f pairs/s family frac %.4f dom %.4f divST %s'%(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dominant_instance']['frac'], d['roofline'].get('divst_subblock',{}).get('ms')))" 2>&1 | tail -1
grep -E "^gemm " $O/step_shapes.txt | head -24 | cut -c1-150
stamp "pretrain_step"
timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/vis.json 2> $O/vis.err
python -c "import json;d=json.load(open('$O/vis.json'));r=d['roofline']['divst_subblock'];print('visual_fwd: %.3f ms  divST %.3f ms frac %.4f (end-to-end %.3f)'%(d['ms_per_step'],r['ms'],r['frac'],r['ms_end_to_end']))" 2>&1 | tail -1
stamp "visual_fwd"
timeout 600 python -m pytest tests/test_model_parity.py tests/test_amp_gpu.py -m gpu -q -p no:cacheprovider --durations=8 > $O/t_model.txt 2>&1; stamp "model tests: $(grep -E 'passed|failed|error' $O/t_model.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/t_model.txt | head
