L=alpro_amd/lib
for i in 1 2; do
  for v in base exp; do
    cp $L/libalpro_hip_$v.so $L/libalpro_hip.so
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('$v', j['ms_per_step'], 'gemm', k['gemm'], 'tn', k['gemm_tn_acc'], 'ln', k['layernorm'], 'lnb', k['layernorm_bwd'], 'gc', k['gather_cast'])"
  done
done
