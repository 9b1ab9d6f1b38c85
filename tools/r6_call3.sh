#!/bin/bash
# round 6, call 3: fused qkv -> temporal attention (numerics + timing), attention-forward variants on one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c3
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "qkv_temporal_attention_fused" ) > $O/pytest_tattn.log 2>&1
tail -15 $O/pytest_tattn.log
timeout 300 python tools/tattn_bench.py > $O/tattn_bench.txt 2>&1
cat $O/tattn_bench.txt
for v in r5 new norot nokv nopk nokvrot r5 new; do
  echo "== $v" >> $O/attn_variants.txt
  ALPRO_HIP_LIB=$R/alpro_amd/lib/variants/libalpro_hip_$v.so ALPRO_BENCH_DTYPE=fp16 timeout 200 python tools/attn_bench.py fwd 2>&1 | grep "vit spatial\|fusion 4B  " >> $O/attn_variants.txt
done
cat $O/attn_variants.txt
for f in 1 0; do
  ALPRO_FUSE_TATTN=$f python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_vfwd_fuse$f.json 2> $O/bench_vfwd_fuse$f.err
done
python - <<'PY'
import json
for f in (1, 0):
    try:
        d=json.loads([l for l in open("gpurun_out/r6c3/bench_vfwd_fuse%d.json" % f) if l.startswith("{")][0])
        print("fuse", f, "vfwd", d["ms_per_step"], d["roofline"]["divst_subblock"]["ms"], d["roofline"]["divst_subblock"]["frac"], d["kernel_ms_per_step"])
    except Exception as e:
        print("fuse", f, "failed", e); print(open("gpurun_out/r6c3/bench_vfwd_fuse%d.err" % f).read()[-1500:])
PY
