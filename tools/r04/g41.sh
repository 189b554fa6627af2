#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python3 -m pytest tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu -k "fwd3_plain_hip_twin" 2>&1 | tail -12 | cut -c1-200
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -3
