#!/bin/bash
# round 5 GPU job: sdpa_bwd_dq2 with the tile's 40 operand reads asked for ahead of its 16 products (DQ2_FAST_TILE, default) against the build without
# (-DDQ2_FAST_TILE=0 -> halva_amd/libhalva_hip_dq2f0.so): kernel tests, bitwise comparison of dq / dk / dv, timing at the step's two launch shapes.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu > $O/r05_dq2fast_pytest.log 2>&1; tail -3 $O/r05_dq2fast_pytest.log
HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_dq2f0.so timeout 300 python3 tools/check_bwd_bitwise.py /tmp/bwd_ab save 2>&1 | tail -1
timeout 300 python3 tools/check_bwd_bitwise.py /tmp/bwd_ab compare 2>&1 | tee $O/r05_dq2fast_bitwise.log | tail -12
VARIANTS="dq2f0 cur dq2f0 cur" SKIPTESTS=1 bash tools/r04/ab_library_variants.sh 2>&1 | grep -E "^==|dq2|dkv3" | tee $O/r05_dq2fast_ab.log
