#!/bin/bash
# per-kernel durations of the SDPA kernels at the step's packed shape (tools/bench_sdpa_branch.py); usage: tools/prof_sdpa_branch.sh <tag>
R=$PWD; tag=${1:-run}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sdpab_$tag -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_sdpab_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"]: print("%-60s calls %4s avg %8.1f us  total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
