#!/bin/bash
# configs[3]/[4] bench lines, the --no-roofline A/B of the default line, the T=4096 packed parity case
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python3 -m pytest tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu -k "vila_row" > $O/r04_pytest_vila_row.log 2>&1; tail -3 $O/r04_pytest_vila_row.log
timeout 900 python3 bench.py --model 13b --no-cpu-baseline > $O/r04_bench_13b.json 2> $O/r04_bench_13b.err; tail -c 1500 $O/r04_bench_13b.json
timeout 900 python3 bench.py --model vila13b --no-cpu-baseline > $O/r04_bench_vila13b.json 2> $O/r04_bench_vila13b.err; tail -c 1500 $O/r04_bench_vila13b.json
timeout 600 python3 bench.py --no-cpu-baseline --no-roofline > $O/r04_bench_noroof.json 2> $O/r04_bench_noroof.err; tail -c 600 $O/r04_bench_noroof.json
timeout 600 python3 bench.py --no-cpu-baseline > $O/r04_bench_roof.json 2> $O/r04_bench_roof.err; tail -c 600 $O/r04_bench_roof.json
