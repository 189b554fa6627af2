#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python3 -m pytest tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu -k "nan_and_inf" 2>&1 | tail -12 | cut -c1-200
