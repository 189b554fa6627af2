#!/bin/bash
# round 5 GPU job: sdpa_fwd3 with the NEXT item's Q fragments requested in the current item's prologue (HALVA_FWD3_QPRE=1, sdpa_fwd3_kernel<true, true>):
# the forward's tests under the switch, then a same-box A/B of the bench line's fwd_in_step.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
HALVA_FWD3_QPRE=1 timeout 600 python3 tools/ab_fwd3.py quick > $O/r05_qpre_ab_fwd3.log 2>&1; tail -25 $O/r05_qpre_ab_fwd3.log
HALVA_FWD3_QPRE=1 timeout 900 python3 -m pytest tests/test_hip_kernels.py tests/test_sdpa_bench_shapes_gpu.py tests/test_hip_graph_capture_gpu.py -x -q -m gpu > $O/r05_qpre_pytest.log 2>&1; tail -5 $O/r05_qpre_pytest.log
HALVA_FWD3_QPRE=1 timeout 300 python3 tools/check_fwd3_det.py > $O/r05_qpre_det.log 2>&1; tail -4 $O/r05_qpre_det.log
for q in 0 1 0 1; do
  HALVA_FWD3_QPRE=$q python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r05_ab_qpre$q.json 2>/dev/null
  python3 - $O/r05_ab_qpre$q.json "qpre=$q" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "pairs/s %.4f ms/step %.1f bwd frac %.4f (%.3f ms) fwd_in_step frac %.4f (%.3f ms)" % (d["value"], d["ms_per_step"], r["frac"], r["launch_ms"], r["fwd_in_step"]["frac"], r["fwd_in_step"]["launch_ms"]))
PY
done
