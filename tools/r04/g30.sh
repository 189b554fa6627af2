#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python3 tools/ab_fwd3.py > $O/r04_ab_fwd3_f.log 2>&1; tail -6 $O/r04_ab_fwd3_f.log
timeout 300 python3 tools/check_fwd3_det.py > $O/r04_det_f.log 2>&1; tail -2 $O/r04_det_f.log
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so STAMP_TAIL=1 timeout 300 python3 tools/stamp_fwd3.py > $O/r04_stamp_fwd3_tail2.log 2>&1; grep "^tail\|^per item" $O/r04_stamp_fwd3_tail2.log
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu > $O/r04_pytest_sdpa_g.log 2>&1; tail -2 $O/r04_pytest_sdpa_g.log
