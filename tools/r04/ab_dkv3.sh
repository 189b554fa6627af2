#!/bin/bash
# time-based A/B of sdpa_bwd_dkv3 at the step's launch shapes: the shipped loop, without its tile requests, without its dS stores, without both,
# without its waits / barriers (gen_dkv3_loop.py DKV3_DIAG builds; results of those are wrong).  rocprofv3 --kernel-trace --stats, one process per variant.
R=$PWD
export BENCH_STEP_SHAPES=1
cd /tmp && export TMPDIR=/tmp
for v in dkvbase dkvnodma dkvnostore dkvnone dkvnobar dkvbase; do
  export HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_$v.so
  rm -rf $R/gpurun_out/prof_dkv3ab_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dkv3ab_$v -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_dkv3ab_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"]: print("  %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
