set -x
timeout 600 python tools/ab_fwd3.py > gpurun_out/r04_ab_fwd3.log 2>&1; echo "rc $?" >> gpurun_out/r04_ab_fwd3.log
HALVA_SDPA_FWD3=0 timeout 900 python -m pytest tests/test_dpa_step_gpu.py -q -s -k "long" > gpurun_out/r04_long_fixture.log 2>&1
tail -30 gpurun_out/r04_ab_fwd3.log
