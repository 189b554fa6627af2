#!/bin/bash
# round 5 GPU job: tests of the changed kernels (fused inverse RoPE, fwd3 repair pass, long-fixture bounds, full-width parity) and a same-box A/B of the bench
# with the inverse RoPE inside the backward epilogues vs as its own launch, and with / without the fwd3 repair pass.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu > $O/r05_pytest_kernels.log 2>&1; tail -3 $O/r05_pytest_kernels.log
timeout 900 python3 -m pytest tests/test_dpa_step_gpu.py tests/test_vila_gpu.py -x -q -m gpu -s > $O/r05_pytest_step.log 2>&1; grep -E "loss err|passed|failed|Error" $O/r05_pytest_step.log | tail -24
timeout 1500 python3 -m pytest tests/test_fullwidth_layer_parity_gpu.py -x -q -m gpu -s --durations=8 > $O/r05_fullwidth_parity.log 2>&1; tail -16 $O/r05_fullwidth_parity.log
for cfg in "1 1" "0 1" "1 0" "1 1" "0 1"; do
  set -- $cfg
  HALVA_ROPE_FUSED_BWD=$1 HALVA_FWD3_REPAIR=$2 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r05_ab_rope$1_repair$2.json 2>/dev/null
  python3 - $O/r05_ab_rope$1_repair$2.json "rope_fused=$1 repair=$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "pairs/s %.4f ms/step %.1f bwd frac %.4f (%.3f ms) fwd_in_step frac %.4f (%.3f ms)" % (d["value"], d["ms_per_step"], r["frac"], r["launch_ms"], r["fwd_in_step"]["frac"], r["fwd_in_step"]["launch_ms"]))
PY
done
