#!/bin/bash
# round 6, call 4: CLS parts by DPP, fused temporal half in training too
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c4
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "qkv_temporal_attention_fused or attn" ) > $O/pytest_ops.log 2>&1
tail -6 $O/pytest_ops.log
timeout 300 python tools/tattn_bench.py > $O/tattn_bench.txt 2>&1
cat $O/tattn_bench.txt
ALPRO_BENCH_DTYPE=fp16 timeout 200 python tools/attn_bench.py fwd 2>&1 | grep "vit spatial\|fusion 4B  " > $O/attn_bench.txt
cat $O/attn_bench.txt
( time timeout 1200 python -m pytest tests/test_model_parity.py -m gpu -x -q ) > $O/pytest_model.log 2>&1
tail -6 $O/pytest_model.log
python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_vfwd.json 2> $O/bench_vfwd.err
for f in 1 0; do
  ALPRO_FUSE_TATTN=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > $O/bench_step_fuse$f.json 2> $O/bench_step_fuse$f.err
done
python - <<'PY'
import json
def rd(f):
    return json.loads([l for l in open("gpurun_out/r6c4/%s.json" % f) if l.startswith("{")][0])
try:
    d = rd("bench_vfwd"); print("vfwd", d["ms_per_step"], d["roofline"]["divst_subblock"]["ms"], d["roofline"]["divst_subblock"]["frac"], d["kernel_ms_per_step"])
except Exception as e:
    print("vfwd failed", e); print(open("gpurun_out/r6c4/bench_vfwd.err").read()[-1500:])
for f in (1, 0):
    try:
        d = rd("bench_step_fuse%d" % f); print("step fuse", f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["kernel_ms_per_step"])
    except Exception as e:
        print("step fuse", f, "failed", e); print(open("gpurun_out/r6c4/bench_step_fuse%d.err" % f).read()[-1500:])
PY
