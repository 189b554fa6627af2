#!/bin/bash
# round 5 GPU job: the frozen reference model's forward on a second stream beside the pair groups (HALVA_OVERLAP_REF=1) against the serial step: the step
# fixture tests under the switch, then an alternating A/B of the bench line.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
HALVA_OVERLAP_REF=1 timeout 900 python3 -m pytest tests/test_dpa_step_gpu.py -x -q -m gpu > $O/r05_overlap_pytest.log 2>&1; tail -3 $O/r05_overlap_pytest.log
for v in 0 1 0 1; do
  HALVA_OVERLAP_REF=$v python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r05_ab_overlap$v.json 2> $O/r05_ab_overlap$v.err || tail -5 $O/r05_ab_overlap$v.err
  python3 - $O/r05_ab_overlap$v.json "overlap_ref=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "pairs/s %.4f ms/step %.1f bwd frac %.4f (%.3f ms) fwd_in_step frac %.4f (%.3f ms) mem %s" % (d["value"], d["ms_per_step"], r["frac"], r["launch_ms"], r["fwd_in_step"]["frac"], r["fwd_in_step"]["launch_ms"], d.get("peak_mem_gib", d["config"].get("peak_mem_gib"))))
PY
done
