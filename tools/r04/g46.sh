#!/bin/bash
cd $GRAFT_REPO_ROOT
HALVA_WGRAD_KT=32 timeout 600 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
timeout 600 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
echo "== KT 64"; timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep "total\|dA qkv\|dB gate \|dB q "
echo "== KT 32"; HALVA_WGRAD_KT=32 timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep "total\|dA qkv\|dB gate \|dB q "
echo "== KT 64"; timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep "total"
echo "== KT 32"; HALVA_WGRAD_KT=32 timeout 300 python3 tools/bench_wgrad_layer.py 2>&1 | grep "total"
