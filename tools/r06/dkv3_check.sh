#!/bin/bash
# round 6: correctness of the SDPA kernels (through the C ABI) + sdpa_bwd_dkv3's anatomy and kernel times for the build in the tree
R=$PWD; mkdir -p gpurun_out
tag=${1:-cur}
timeout 1500 python3 -m pytest tests/test_sdpa_bench_shapes_gpu.py tests/test_hip_kernels.py tests/test_hip_graph_capture_gpu.py -x -q -m gpu > gpurun_out/r06_pytest_sdpa_$tag.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r06_pytest_sdpa_$tag.log
{
  if [ -f $R/halva_amd/libhalva_hip_stamp2.so ]; then
    echo "== stamp build (HALVA_STAMP, stride 1), S=16 plain rows of 2048"
    S=16 HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_stamp2.so timeout 300 python3 tools/stamp_persistent.py
  fi
  echo "== build in the tree, step shapes"
  BENCH_STEP_SHAPES=1 timeout 300 python3 tools/bench_sdpa_branch.py
} > gpurun_out/r06_dkv3_anatomy_$tag.log 2>&1
cd /tmp && export TMPDIR=/tmp
export BENCH_STEP_SHAPES=1
rm -rf $R/gpurun_out/prof_r06$tag
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06$tag -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_r06$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" >> $R/gpurun_out/r06_dkv3_anatomy_$tag.log <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"] or "rope" in r["Name"]: print("  %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat $R/gpurun_out/r06_dkv3_anatomy_$tag.log
