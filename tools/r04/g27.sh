#!/bin/bash
# full GPU suite + bench A/B of the LDS-DMA weight-gradient kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/r04_pytest_gpu_5.log 2>&1; tail -4 $O/r04_pytest_gpu_5.log
HALVA_WGRAD_DMA=0 timeout 600 python3 bench.py --no-cpu-baseline > $O/r04_bench_wgrad0.json 2> $O/r04_bench_wgrad0.err
timeout 600 python3 bench.py --no-cpu-baseline > $O/r04_bench_f.json 2> $O/r04_bench_f.err
grep -h -o '"value": [0-9.]*, "unit": "paired-samples/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/r04_bench_wgrad0.json $O/r04_bench_f.json
grep -o '"frac": [0-9.]*, "traffic"\|"fwd_in_step": {"achieved": [0-9.]*, "frac": [0-9.]*' $O/r04_bench_f.json
