#!/bin/bash
# round 6, first GPU call: sdpa_bwd_dkv3's per-item anatomy re-collected (the r04 log predates the fused rotation) + the kernels' times at the step's two shapes
R=$PWD; mkdir -p gpurun_out
{
  echo "== stamp build (HALVA_STAMP, stride 1), S=16 plain rows of 2048"
  S=16 HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_stamp1.so python3 tools/stamp_persistent.py
  echo "== shipped build, step shapes"
  BENCH_STEP_SHAPES=1 python3 tools/bench_sdpa_branch.py
} > gpurun_out/r06_dkv3_anatomy_base.log 2>&1
cd /tmp && export TMPDIR=/tmp
export BENCH_STEP_SHAPES=1
rm -rf $R/gpurun_out/prof_r06base
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06base -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_r06base -name '*kernel_stats.csv' | head -1)
python3 - "$f" >> $R/gpurun_out/r06_dkv3_anatomy_base.log <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"] or "rope" in r["Name"]: print("  %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat $R/gpurun_out/r06_dkv3_anatomy_base.log
