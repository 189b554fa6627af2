#!/bin/bash
# sdpa_bwd_dq2 "anatomy by removal" (round 5): builds of the product library with -DHALVA_DQ2_DIAG=<bits> (sdpa.hip: 1 no matrix work, 2 no dS requests,
# 4 no K requests, 8 no barriers) -> halva_amd/libhalva_hip_dq2d<bits>.so (git-ignored; they travel with gpurun).  Timing only: the results are wrong.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/halva_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for bits in ${BITS:-1 2 4 8 3}; do
  mkdir -p /tmp/dq2d_$bits
  make -j8 OBJDIR=/tmp/dq2d_$bits OUT=$R/halva_amd/libhalva_hip_dq2d$bits.so CXXFLAGS="$BASE -DHALVA_DQ2_DIAG=$bits" 2>&1 | grep -E "rror|FAILED" || true
  ls -la $R/halva_amd/libhalva_hip_dq2d$bits.so
done
