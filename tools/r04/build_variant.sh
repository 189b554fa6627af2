#!/bin/bash
# usage: build_variant.sh <name> [ENV=VAL ...]  -> halva_amd/libhalva_hip_<name>.so (a -DHALVA_STAMP build with the generator run under the given environment)
set -e
name=$1; shift
cd /root/repo/halva_amd/csrc
env "$@" python3 gen_fwd3_loop.py > /dev/null
mkdir -p /tmp/st_$name
make -j8 OBJDIR=/tmp/st_$name OUT=/root/repo/halva_amd/libhalva_hip_$name.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DHALVA_STAMP" 2>&1 | grep -E "rror|FAILED" || true
python3 gen_fwd3_loop.py > /dev/null
ls -la /root/repo/halva_amd/libhalva_hip_$name.so
