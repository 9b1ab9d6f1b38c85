#!/bin/bash
# Same-box A/B of two builds of the library under the default bench (run through gpurun from the repo root):
#   build the baseline, cp alpro_amd/lib/libalpro_hip.so alpro_amd/lib/libalpro_hip_base.so; edit + rebuild,
#   cp alpro_amd/lib/libalpro_hip.so alpro_amd/lib/libalpro_hip_exp.so; gpurun -- 'bash tools/ab_lib.sh'
# The two libraries are swapped in turn, twice each; afterwards libalpro_hip.so is the "exp" build -- rebuild (python -m alpro_amd.build)
# or copy the one you keep.  Boxes of the pool differ by up to 7 % on one binary, so only same-box numbers compare.
L=alpro_amd/lib
for i in 1 2; do
  for v in base exp; do
    cp $L/libalpro_hip_$v.so $L/libalpro_hip.so
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('$v', j['ms_per_step'], 'gemm', k['gemm'], 'tn', k['gemm_tn_acc'], 'ln', k['layernorm'], 'lnb', k['layernorm_bwd'], 'gc', k['gather_cast'], 'attn', k['attn'], k['attn_temporal'], 'attnb', k['attn_bwd'])"
  done
done
