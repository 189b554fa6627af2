set -x
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python tools/stamp_fwd3.py > gpurun_out/r04_stamp_fwd3.log 2>&1
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python tools/stamp_fwd3.py packed > gpurun_out/r04_stamp_fwd3_packed.log 2>&1
tail -40 gpurun_out/r04_stamp_fwd3.log
