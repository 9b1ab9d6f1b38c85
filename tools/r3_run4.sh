set -x
mkdir -p gpurun_out/r3d
python -m pytest tests/test_hip_bwd_ops.py tests/test_hip_ops.py tests/test_input_gpu.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r3d/tests_ops.txt
python -m pytest tests/test_model_parity.py -m gpu -q -k "gradients or released" 2>&1 | tail -30 > gpurun_out/r3d/tests_model.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d/bench_fp16.json 2> gpurun_out/r3d/bench_fp16.err
ALPRO_SAVE_GELU_GRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-divst > gpurun_out/r3d/bench_fp16_nogelugrad.json 2> gpurun_out/r3d/bench_fp16_nogelugrad.err
ALPRO_BENCH_SHAPES=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-divst > gpurun_out/r3d/bench_shapes.json 2> gpurun_out/r3d/bench_shapes.err
python tools/aten_tail.py > gpurun_out/r3d/aten_tail.txt 2>&1
tail -n 4 gpurun_out/r3d/tests_ops.txt gpurun_out/r3d/tests_model.txt
