#!/bin/bash
# round 6, last GPU job: smoke, the whole GPU suite, then everything profiles/ holds for the round from the final build, then the natural-length line
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/r06_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r06_smoke.log
timeout 3000 python3 -m pytest tests -q -m gpu > $O/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r06_pytest_gpu.log
HALVA_COMMIT=$1 bash tools/collect_round_profiles.sh r06 > $O/r06_collect.log 2>&1; tail -42 $O/r06_collect.log | head -36
python3 bench.py --steps 20 --warmup 3 --resp-len 128 --pairs-per-gpu 8 --no-cpu-baseline > $O/r06_bench_natural.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/r06_bench_natural.json'));print('natural', d['value'], d['ms_per_step'])"
