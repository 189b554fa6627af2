#!/bin/bash
# fwd3 with the row sums as v_add_f32 / CAP 6: correctness + timing against the old kernel, determinism, SDPA tests, bench
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python3 tools/ab_fwd3.py > $O/r04_ab_fwd3_e.log 2>&1; tail -25 $O/r04_ab_fwd3_e.log
timeout 300 python3 tools/check_fwd3_det.py > $O/r04_det_e.log 2>&1; tail -4 $O/r04_det_e.log
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu > $O/r04_pytest_sdpa_e.log 2>&1; tail -3 $O/r04_pytest_sdpa_e.log
timeout 600 python3 bench.py --no-cpu-baseline > $O/r04_bench_e.json 2> $O/r04_bench_e.err; grep -o '"value": [0-9.]*\|"fwd_in_step": {[^}]*}' $O/r04_bench_e.json | head -3
