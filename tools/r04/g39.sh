#!/bin/bash
cd $GRAFT_REPO_ROOT; timeout 600 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad" 2>&1 | tail -4
