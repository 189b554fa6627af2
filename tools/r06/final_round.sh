#!/bin/bash
# round 6: the tests that changed since the last full run, then everything profiles/ holds for the round from the final build
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_dp_engine_gpu.py tests/test_fullwidth_layer_parity_gpu.py tests/test_train_e2e_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_hip_kernels.py tests/test_dpa_step_gpu.py -q -m gpu -s > $O/r06_pytest_changed.log 2>&1
echo "pytest rc=$?"; tail -8 $O/r06_pytest_changed.log
HALVA_COMMIT=$1 bash tools/collect_round_profiles.sh r06 > $O/r06_collect.log 2>&1; tail -45 $O/r06_collect.log
