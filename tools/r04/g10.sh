timeout 600 python tools/ab_fwd3.py > gpurun_out/r04_ab_fwd3_d.log 2>&1; tail -22 gpurun_out/r04_ab_fwd3_d.log
timeout 600 python tools/check_fwd3_det.py > gpurun_out/r04_det_d.log 2>&1; grep -v "rows beyond" gpurun_out/r04_det_d.log | tail -10
HALVA_HIP_LIB=$PWD/halva_amd/libhalva_hip_stamp.so timeout 300 python tools/stamp_fwd3.py > gpurun_out/r04_stamp_fwd3_d.log 2>&1; sed -n 2,3p gpurun_out/r04_stamp_fwd3_d.log; sed -n 22,30p gpurun_out/r04_stamp_fwd3_d.log; tail -13 gpurun_out/r04_stamp_fwd3_d.log | head -4
