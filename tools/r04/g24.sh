#!/bin/bash
cd $GRAFT_REPO_ROOT
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python3 tools/stamp_fwd3.py > gpurun_out/r04_stamp_fwd3_e.log 2>&1; cat gpurun_out/r04_stamp_fwd3_e.log | tail -60
