#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "kl_rows" 2>&1 | tail -3
timeout 300 python3 tools/bench_kl_rows.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_bench_kl_rows.log
