python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu_2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_pytest_gpu_2.log
tail -15 gpurun_out/r04_pytest_gpu_2.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_fwd3_a.json 2> gpurun_out/r04_bench_fwd3_a.err
HALVA_SDPA_FWD3=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_fwd3_off.json 2> gpurun_out/r04_bench_fwd3_off.err
