set -x
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -q --ignore=tests/test_model_parity.py 2>&1 | tail -30 > gpurun_out/r3a/tests.txt
bash tools/parity_report.sh gpurun_out/r3a/parity_report.txt
python bench.py --steps 10 --warmup 3 > gpurun_out/r3a/bench_fp16.json 2> gpurun_out/r3a/bench_fp16.err
python bench.py --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline > gpurun_out/r3a/bench_bf16.json 2> gpurun_out/r3a/bench_bf16.err
tail -5 gpurun_out/r3a/tests.txt; tail -3 gpurun_out/r3a/bench_fp16.err; cat gpurun_out/r3a/bench_fp16.json | cut -c1-1500
