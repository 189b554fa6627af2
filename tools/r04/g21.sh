#!/bin/bash
cd $GRAFT_REPO_ROOT; timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_bench_wgrad.log
