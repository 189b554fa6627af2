#!/bin/bash
# top-layer row pruning: tests, then A/B of the default line; configs[3]/[4] bench lines; the --no-roofline A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python3 -m pytest tests/test_residual_inplace_gpu.py tests/test_dpa_step_gpu.py tests/test_vila_gpu.py tests/test_dp_engine_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu > $O/r04_pytest_toprows.log 2>&1; tail -5 $O/r04_pytest_toprows.log
HALVA_TOP_ROWS=0 timeout 600 python3 bench.py --no-cpu-baseline > $O/r04_bench_toprows0.json 2> $O/r04_bench_toprows0.err; tail -c 300 $O/r04_bench_toprows0.err
timeout 600 python3 bench.py --no-cpu-baseline > $O/r04_bench_roof.json 2> $O/r04_bench_roof.err; tail -c 300 $O/r04_bench_roof.err
timeout 600 python3 bench.py --no-cpu-baseline --no-roofline > $O/r04_bench_noroof.json 2> $O/r04_bench_noroof.err
timeout 900 python3 bench.py --model 13b --no-cpu-baseline > $O/r04_bench_13b.json 2> $O/r04_bench_13b.err; tail -c 300 $O/r04_bench_13b.err
timeout 900 python3 bench.py --model vila13b --no-cpu-baseline > $O/r04_bench_vila13b.json 2> $O/r04_bench_vila13b.err; tail -c 300 $O/r04_bench_vila13b.err
grep -h -o '"value": [0-9.]*, "unit": "paired-samples/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/r04_bench_toprows0.json $O/r04_bench_roof.json $O/r04_bench_noroof.json $O/r04_bench_13b.json $O/r04_bench_vila13b.json
