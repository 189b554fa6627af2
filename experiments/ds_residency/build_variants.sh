#!/bin/bash
# experiments/ds_residency: two builds of the product library with the eviction hook (-DHALVA_DS_EVICT_EXP) between sdpa_bwd_dkv3 and sdpa_bwd_dq2:
#   dsres_nt     dS stored and fetched nontemporally (what ships)
#   dsres_plain  default cache policy on both sides
# -> halva_amd/libhalva_hip_dsres_{nt,plain}.so (git-ignored; they travel with gpurun)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/halva_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DHALVA_DS_EVICT_EXP"
build() {      # name, generator policy, extra flags
  DKV3_DS_POLICY=$2 python3 gen_dkv3_loop.py > /dev/null
  mkdir -p /tmp/dsres_$1
  make -j8 OBJDIR=/tmp/dsres_$1 OUT=$R/halva_amd/libhalva_hip_dsres_$1.so CXXFLAGS="$BASE $3" 2>&1 | grep -E "rror|FAILED" || true
  ls -la $R/halva_amd/libhalva_hip_dsres_$1.so
}
build nt nt ""
build plain plain "-DHALVA_DQ2_DS_POLICY='\"\"'"
python3 gen_dkv3_loop.py > /dev/null
