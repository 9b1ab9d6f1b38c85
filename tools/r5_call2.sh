#!/bin/bash
# round 5, GPU call 2 (scheduler v2: static pair + claim words + scan stealing): the dynamic tile scheduler (tests, kernel-level A/B under CU theft, step-level contention table), the side-stream CLS chain
# (neutrality test, A/B on both bench workloads), full-size backward parity (first measurement).
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 120 python tools/sched_smoke.py > $O/smoke.txt 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.txt; exit 1; }
stamp "smoke: $(tail -1 $O/smoke.txt)"
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -x -k "8phase or scheduler or persistent" --durations=12 > $O/t_sched.txt 2>&1; stamp "sched tests: $(grep -E 'passed|failed|error' $O/t_sched.txt | tail -1)"
timeout 600 python -m pytest tests/test_model_parity.py -m gpu -q -p no:cacheprovider -s -k "full_size_pretrain_backward" --durations=5 > $O/t_model.txt 2>&1; stamp "model tests: $(grep -E 'passed|failed|error' $O/t_model.txt | tail -1)"; grep -E "B=64 backward" $O/t_model.txt | cut -c1-900
timeout 300 python tools/sched_contention.py > $O/sched_contention.txt 2>&1; stamp "sched_contention"; cat $O/sched_contention.txt | cut -c1-200
for cs in 0 1; do for sc in 1 0; do
  ALPRO_CLS_STREAM=$cs ALPRO_GEMM_SCHED=$sc timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/vis_cs${cs}_sc${sc}.json 2> $O/vis_cs${cs}_sc${sc}.err
  python -c "import json;d=json.load(open('$O/vis_cs${cs}_sc${sc}.json'));r=d['roofline']['divst_subblock'];print('visual_fwd cls_stream=$cs sched=$sc: %.3f ms  divST %.3f ms frac %.4f (end-to-end %.3f)'%(d['ms_per_step'],r['ms'],r['frac'],r['ms_end_to_end']))" 2>&1 | tail -1
done; done
stamp "visual_fwd A/B"
for cs in 0 1; do
  ALPRO_CLS_STREAM=$cs timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_cs${cs}.json 2> $O/step_cs${cs}.err
  python -c "import json;d=json.load(open('$O/step_cs${cs}.json'));print('pretrain_step cls_stream=$cs: %.3f ms %.1f pairs/s family frac %.4f dom %.4f'%(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dominant_instance']['frac']))" 2>&1 | tail -1
done
ALPRO_GEMM_SCHED=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/step_sc0.json 2> $O/step_sc0.err
python -c "import json;d=json.load(open('$O/step_sc0.json'));print('pretrain_step sched=0: %.3f ms %.1f pairs/s family frac %.4f dom %.4f'%(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dominant_instance']['frac']))" 2>&1 | tail -1
stamp "pretrain_step A/B"
timeout 400 python tools/overlap_contention.py --steps 3 > $O/overlap_contention.txt 2>&1; stamp "overlap_contention"; tail -22 $O/overlap_contention.txt
