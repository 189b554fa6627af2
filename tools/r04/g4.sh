set -x
timeout 600 python tools/ab_fwd3.py > gpurun_out/r04_ab_fwd3_b.log 2>&1; echo "rc $?" >> gpurun_out/r04_ab_fwd3_b.log
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python tools/stamp_fwd3.py > gpurun_out/r04_stamp_fwd3_b.log 2>&1
tail -30 gpurun_out/r04_ab_fwd3_b.log
tail -24 gpurun_out/r04_stamp_fwd3_b.log
