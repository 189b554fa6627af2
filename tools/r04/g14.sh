python -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py tests/test_cabi_symbols.py -m "gpu or not gpu" -q -k "sdpa or dkv3 or symbols or sync" > gpurun_out/r04_pytest_sdpa.log 2>&1; tail -5 gpurun_out/r04_pytest_sdpa.log
R=$PWD; export BENCH_STEP_SHAPES=1; cd /tmp; export TMPDIR=/tmp
for v in dkvbase dkvlean dkvbase dkvlean; do
  export HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_$v.so
  rm -rf $R/gpurun_out/prof_dkv3ab_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dkv3ab_$v -o p --output-format csv -- python3 $R/tools/bench_sdpa_branch.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_dkv3ab_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sdpa" in r["Name"]: print("  %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
