#!/bin/bash
cd $GRAFT_REPO_ROOT
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so STAMP_TAIL=1 timeout 300 python3 tools/stamp_fwd3.py > gpurun_out/r04_stamp_fwd3_tail.log 2>&1; grep "^tail\|^per item" gpurun_out/r04_stamp_fwd3_tail.log
