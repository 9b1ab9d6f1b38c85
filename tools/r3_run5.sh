set -x
mkdir -p gpurun_out/r3e
python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r3e/tests_ops.txt
ALPRO_BENCH_SHAPES=1 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r3e/bench_visual_tail1.json 2> gpurun_out/r3e/bench_visual_tail1.err
ALPRO_GEMM_TAIL=0 ALPRO_BENCH_SHAPES=1 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r3e/bench_visual_tail0.json 2> gpurun_out/r3e/bench_visual_tail0.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > gpurun_out/r3e/bench_step_tail1.json 2> gpurun_out/r3e/bench_step_tail1.err
ALPRO_GEMM_TAIL=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > gpurun_out/r3e/bench_step_tail0.json 2> gpurun_out/r3e/bench_step_tail0.err
tail -n 3 gpurun_out/r3e/tests_ops.txt
