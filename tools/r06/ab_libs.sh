#!/bin/bash
# usage: ab_libs.sh <rounds> <lib suffix> ...   - the SDPA kernels' average times at the step's two launch shapes (tools/bench_sdpa_branch.py under
# rocprofv3 --kernel-trace --stats) for several builds of the library, alternating on ONE box
R=$PWD; mkdir -p gpurun_out; rounds=$1; shift
cd /tmp && export TMPDIR=/tmp
export BENCH_STEP_SHAPES=1
for r in $(seq $rounds); do for v in "$@"; do
  export HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_$v.so
  rm -rf /tmp/prof_ab_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab_$v -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
  f=$(find /tmp/prof_ab_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
t = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("dkv3", "dq2", "delta", "fwd3"):
        if "sdpa_" in r["Name"] and k in r["Name"]: t[k] = float(r["AverageNs"]) / 1e3
print("%-8s dkv3 %8.1f  dq2 %7.1f  delta %6.1f  fwd3 %7.1f us" % (sys.argv[2], t.get("dkv3", 0), t.get("dq2", 0), t.get("delta", 0), t.get("fwd3", 0)))
PY
done; done | tee $R/gpurun_out/r06_ab_libs.log
