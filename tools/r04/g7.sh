timeout 900 python tools/check_fwd3_det.py > gpurun_out/r04_det.log 2>&1; echo "rc $?" >> gpurun_out/r04_det.log
cat gpurun_out/r04_det.log
timeout 600 python tools/ab_fwd3.py quick > gpurun_out/r04_ab_fwd3_c.log 2>&1; tail -16 gpurun_out/r04_ab_fwd3_c.log
