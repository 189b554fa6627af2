#!/bin/bash
# round 4: does a deeper operand look-ahead help the dK/dV step?  (run from this directory; ./step_bench_<name> 2048 1024)
rm -f step_bench_*
build() { name=$1; shift; env "$@" python3 gen_step_asm.py > /dev/null && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -w -o step_bench_$name step_bench.hip; }
build la6
build la7 DKV_LA=7
build la10_r16 DKV_LA=10 DKV_NRING=16
build la14_r16 DKV_LA=14 DKV_NRING=16
build la6_c6 DKV_CAP=6
python3 gen_step_asm.py > /dev/null
