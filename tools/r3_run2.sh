set -x
mkdir -p gpurun_out/r3b
python -m pytest tests/test_hip_ops.py tests/test_amp_gpu.py tests/test_dist_gpu.py tests/test_hip_bwd_ops.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r3b/tests_ops.txt
python -m pytest tests/test_model_parity.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r3b/tests_model.txt
bash tools/parity_report.sh gpurun_out/r3b/parity_report.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3b/bench_fp16_fused.json 2> gpurun_out/r3b/bench_fp16_fused.err
ALPRO_FUSE_RESIDUAL_LN=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3b/bench_fp16_unfused.json 2> gpurun_out/r3b/bench_fp16_unfused.err
ALPRO_BENCH_SHAPES=1 python bench.py --workload visual_fwd --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3b/bench_visual_fused.json 2> gpurun_out/r3b/bench_visual_fused.err
tail -4 gpurun_out/r3b/tests_ops.txt gpurun_out/r3b/tests_model.txt
