#!/bin/bash
cd $GRAFT_REPO_ROOT
S=16 HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python3 tools/stamp_persistent.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_stamp_dkv3.log
