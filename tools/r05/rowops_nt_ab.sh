#!/bin/bash
# round 5 GPU job: same-box A/B of the bench with the streaming row kernels / wgrad / delta loads before (libhalva_hip_rowold.so = the build at commit
# 3b478bc+) and after the nontemporal + one-chunk-per-thread change; then the kernel tests.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in old new old new; do
  if [ $v = old ]; then export HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_rowold.so; else unset HALVA_HIP_LIB; fi
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r05_ab_rows_$v.json 2>/dev/null
  python3 - $O/r05_ab_rows_$v.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "pairs/s %.4f ms/step %.1f loss %s bwd frac %.4f (%.3f ms) fwd_in_step frac %.4f" % (d["value"], d["ms_per_step"], d.get("loss"), r["frac"], r["launch_ms"], r["fwd_in_step"]["frac"]))
PY
done 2>&1 | tee $O/r05_ab_rows.log
unset HALVA_HIP_LIB
timeout 1200 python3 -m pytest tests/test_hip_kernels.py tests/test_dpa_step_gpu.py tests/test_residual_inplace_gpu.py -x -q -m gpu 2>&1 | tail -3
