set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu_1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_pytest_gpu_1.log
HALVA_DP_FORCE=1 HALVA_DIST_BACKEND=nccl python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_dp1_rccl.json 2> gpurun_out/r04_bench_dp1_rccl.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_dp1_plain.json 2> gpurun_out/r04_bench_dp1_plain.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --pairs-per-gpu 8 > gpurun_out/r04_bench_den_8.json 2> gpurun_out/r04_bench_den_8.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --pairs-per-gpu 16 --pairs-per-group 8 > gpurun_out/r04_bench_den_16g8.json 2> gpurun_out/r04_bench_den_16g8.err
python bench.py --steps 3 --warmup 1 --no-roofline > gpurun_out/r04_bench_cpu.json 2> gpurun_out/r04_bench_cpu.err
tail -3 gpurun_out/r04_pytest_gpu_1.log
