#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/r04_pytest_gpu_6.log 2>&1; tail -3 $O/r04_pytest_gpu_6.log
