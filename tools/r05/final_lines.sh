#!/bin/bash
# round 5 GPU job: the HIP-graph capture test, the default bench line, and the configs[3] / configs[4] geometries on one GPU
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python3 -m pytest tests/test_hip_graph_capture_gpu.py -x -q -m gpu > $O/r05_pytest_graph.log 2>&1; tail -5 $O/r05_pytest_graph.log
python3 bench.py --steps 10 --warmup 2 > $O/r05_bench_default.json 2> $O/r05_bench_default.err; tail -c 2500 $O/r05_bench_default.json
python3 bench.py --model 13b --steps 4 --warmup 1 --no-cpu-baseline > $O/r05_bench_13b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/r05_bench_13b.json'));print('13b', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['fwd_in_step']['frac'])"
python3 bench.py --model vila13b --steps 4 --warmup 1 --no-cpu-baseline > $O/r05_bench_vila13b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/r05_bench_vila13b.json'));print('vila13b', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['fwd_in_step']['frac'])"
