#!/bin/bash
# round 6: (1) the bf16 floor of the 8-layer full-width case (tests/golden/fulldepth8_floor.json), (2) the whole GPU test suite
R=$PWD; mkdir -p gpurun_out
timeout 1500 python3 tools/fulldepth_parity.py --layers 8 --resp-len 203 --image 224 --clip-layers 2 --seed 77 \
    --reals plain,perm1,perm2,perm5,perm6,perm7,perm8 --write-floor gpurun_out/fulldepth8_floor.json > gpurun_out/r06_fulldepth8_floor.log 2>&1
echo "floor rc=$?"; grep -v amdgpu.ids gpurun_out/r06_fulldepth8_floor.log | tail -14
cp gpurun_out/fulldepth8_floor.json tests/golden/fulldepth8_floor.json
timeout 3000 python3 -m pytest tests -q -m gpu -s > gpurun_out/r06_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r06_pytest_gpu.log
