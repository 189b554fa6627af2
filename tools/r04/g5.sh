HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python tools/stamp_fwd3.py > gpurun_out/r04_stamp_fwd3_c.log 2>&1
sed -n 1,3p gpurun_out/r04_stamp_fwd3_c.log; sed -n 20,40p gpurun_out/r04_stamp_fwd3_c.log; tail -14 gpurun_out/r04_stamp_fwd3_c.log
