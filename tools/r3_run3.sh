set -x
mkdir -p gpurun_out/r3c
python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py tests/test_amp_gpu.py tests/test_dist_gpu.py tests/test_input_gpu.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r3c/tests_ops.txt
python -m pytest tests/test_model_parity.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r3c/tests_model.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c/bench_fp16.json 2> gpurun_out/r3c/bench_fp16.err
ALPRO_FUSE_LN_BWD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > gpurun_out/r3c/bench_fp16_nolnbwd.json 2> gpurun_out/r3c/bench_fp16_nolnbwd.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst --dtype bf16 > gpurun_out/r3c/bench_bf16.json 2> gpurun_out/r3c/bench_bf16.err
python tools/matmul_probe.py > gpurun_out/r3c/matmul_probe.txt 2>&1
tail -n 4 gpurun_out/r3c/tests_ops.txt gpurun_out/r3c/tests_model.txt
