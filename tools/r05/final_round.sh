#!/bin/bash
# round 5, last GPU job: the whole GPU suite, then everything profiles/ holds for the round from the final build (tools/collect_round_profiles.sh), then the
# configs[3] / configs[4] geometries on one GPU
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/r05_pytest_gpu.log 2>&1; tail -3 $O/r05_pytest_gpu.log
HALVA_COMMIT=$1 bash tools/collect_round_profiles.sh r05 > $O/r05_collect.log 2>&1; tail -45 $O/r05_collect.log
python3 bench.py --model 13b --steps 4 --warmup 1 --no-cpu-baseline > $O/r05_bench_13b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/r05_bench_13b.json'));print('13b', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['fwd_in_step']['frac'])"
python3 bench.py --model vila13b --steps 4 --warmup 1 --no-cpu-baseline > $O/r05_bench_vila13b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/r05_bench_vila13b.json'));print('vila13b', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['fwd_in_step']['frac'])"
