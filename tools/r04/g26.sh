#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad or weight" 2>&1 | tail -4
echo "== LDS-DMA kernel"; timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_bench_wgrad_dma.log
echo "== HALVA_WGRAD_DMA=0"; HALVA_WGRAD_DMA=0 timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep -v amdgpu.ids | grep "total\|dA qkv\|dB gate " 
