#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/r04_pytest_gpu_7.log 2>&1; tail -3 $O/r04_pytest_gpu_7.log
bash tools/r04/g15.sh
