#!/bin/bash
export VARIANTS="dqold cur dqold cur"
bash tools/r04/g20.sh 2>&1 | grep -v "dkv3\|fwd3\|delta"
cd $GRAFT_REPO_ROOT; HALVA_SDPA_SLOW_TR=1 timeout 600 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "sdpa" 2>&1 | tail -2
