#!/bin/bash
# round 5 GPU job: tests of the changed kernels, then same-box A/Bs: sdpa_bwd_dq3 (dS three tiles ahead through registers) vs sdpa_bwd_dq2, and the inverse
# RoPE inside the backward epilogues vs as its own launch - per kernel at the step's two launch shapes (rocprofv3 --kernel-trace --stats) and as bench lines.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu > $O/r05_pytest_kernels.log 2>&1; tail -3 $O/r05_pytest_kernels.log
timeout 900 python3 -m pytest tests/test_dpa_step_gpu.py tests/test_vila_gpu.py -x -q -m gpu -s > $O/r05_pytest_step.log 2>&1; grep -E "loss err|passed|failed|Error" $O/r05_pytest_step.log | tail -24
timeout 1500 python3 -m pytest tests/test_fullwidth_layer_parity_gpu.py -x -q -m gpu -s --durations=8 > $O/r05_fullwidth_parity.log 2>&1; tail -16 $O/r05_fullwidth_parity.log
export BENCH_STEP_SHAPES=1
cd /tmp && export TMPDIR=/tmp
for cfg in "1" "0" "1" "0"; do
  export HALVA_SDPA_DQ3=$cfg
  rm -rf /tmp/prof_dq
  rocprofv3 --kernel-trace --stats -d /tmp/prof_dq -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /tmp/bsb.log 2>&1
  f=$(find /tmp/prof_dq -name '*kernel_stats.csv' | head -1)
  echo "== HALVA_SDPA_DQ3=$cfg"; grep -E "fwd|bwd" /tmp/bsb.log
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"]: print("  %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $O/r05_dq3_ab.log
unset HALVA_SDPA_DQ3
cd $R
for cfg in "1 1" "0 1" "1 0" "1 1" "0 1" "1 0"; do
  set -- $cfg
  HALVA_ROPE_FUSED_BWD=$1 HALVA_SDPA_DQ3=$2 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r05_ab_rope$1_dq3$2.json 2>/dev/null
  python3 - $O/r05_ab_rope$1_dq3$2.json "rope_fused=$1 dq3=$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "pairs/s %.4f ms/step %.1f loss %s bwd frac %.4f (%.3f ms) fwd_in_step frac %.4f (%.3f ms)" % (d["value"], d["ms_per_step"], d.get("loss"), r["frac"], r["launch_ms"], r["fwd_in_step"]["frac"], r["fwd_in_step"]["launch_ms"]))
PY
done 2>&1 | tee $O/r05_ab_rope_dq3.log
