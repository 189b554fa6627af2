#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -3
export VARIANTS="prev cur prev cur" SKIPTESTS=1
bash tools/r04/g20.sh 2>&1 | grep "==\|dkv3\|delta"
