#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python3 bench.py --model 13b --no-cpu-baseline > $O/r04_bench_13b.json 2> $O/r04_bench_13b.err
timeout 900 python3 bench.py --model vila13b --no-cpu-baseline > $O/r04_bench_vila13b.json 2> $O/r04_bench_vila13b.err
grep -h -o '"value": [0-9.]*, "unit": "paired-samples/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/r04_bench_13b.json $O/r04_bench_vila13b.json
