#!/bin/bash
# usage: stamp_variants.sh <lib suffix> ...   - sdpa_bwd_dkv3's per-item anatomy (tools/stamp_persistent.py, S=16 plain rows of 2048) for several -DHALVA_STAMP builds
R=$PWD; mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  S=16 HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_$v.so timeout 300 python3 tools/stamp_persistent.py 2>&1 | grep -v amdgpu.ids | head -4
done > gpurun_out/r06_stamp_variants.log 2>&1
cat gpurun_out/r06_stamp_variants.log
