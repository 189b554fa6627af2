#!/bin/bash
# usage: build_dkv3_variant.sh <name> [ENV=VAL ...]  -> halva_amd/libhalva_hip_<name>.so (an ordinary build with gen_dkv3_loop.py run under the given environment,
# e.g. DKV3_DIAG=nodma,nostore: timing experiments, results wrong)
set -e
name=$1; shift
cd /root/repo/halva_amd/csrc
env "$@" python3 gen_dkv3_loop.py > /dev/null
mkdir -p /tmp/dv_$name
make -j8 OBJDIR=/tmp/dv_$name OUT=/root/repo/halva_amd/libhalva_hip_$name.so 2>&1 | grep -E "rror|FAILED" || true
python3 gen_dkv3_loop.py > /dev/null
ls -la /root/repo/halva_amd/libhalva_hip_$name.so
