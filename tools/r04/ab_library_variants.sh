#!/bin/bash
# Time-based A/B of library variants (halva_amd/libhalva_hip_<name>.so from build_dkv3_variant.sh / build_variant.sh; "cur" = the in-tree build) at the
# step's two SDPA launch shapes, rocprofv3 --kernel-trace --stats, one process per variant:   VARIANTS="cur nostat nodma cur" SKIPTESTS=1 bash tools/r04/ab_library_variants.sh
# (round 4 ran it for the merged LDS waits - VARIANTS "w0 cur w0 cur" - and for the cost of sdpa_bwd_dkv3's statistics / tile requests: DESIGN.md 5.2)
cd $GRAFT_REPO_ROOT; O=gpurun_out; R=$PWD
[ -n "$SKIPTESTS" ] || timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py -x -q -m gpu > $O/r04_pytest_sdpa_f.log 2>&1; tail -3 $O/r04_pytest_sdpa_f.log
export BENCH_STEP_SHAPES=1
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-w0 cur w0 cur}; do
  if [ $v = cur ]; then unset HALVA_HIP_LIB; else export HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_$v.so; fi
  rm -rf $R/gpurun_out/prof_ab_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab_$v -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_ab_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"]: print("  %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $R/gpurun_out/prof_ab_$v
done
