#!/bin/bash
# Collects the judged profile artefacts of the current build into gpurun_out/<tag>/ (run on the GPU box: gpurun -- tools/refresh_profiles.sh <tag>):
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`
#   sq/summary.json    SQ counters of the SDPA kernels (tools/pmc_sdpa.sh)
#   traffic.json       FETCH_SIZE / WRITE_SIZE of the SDPA kernels, separate passes (microbench shape); traffic_step.json: in the bench step
#   bench_stdout.log   the default `python bench.py` line
tag=${1:-final}; R=$PWD; OUT=$R/gpurun_out/$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_profiled_stdout.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $OUT/traffic/$c -o t --output-format csv -- python3 $R/tools/bench_sdpa.py > /dev/null 2>&1)
done
python3 tools/summarize_pmc.py $OUT/traffic sdpa_fwd sdpa_bwd_dq sdpa_bwd_dkv sdpa_bwd_delta > $OUT/traffic.json
# the same two counters on the launches of the bench step itself (16-sequence groups, packed rows)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $OUT/traffic_step/$c -o t --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
done
python3 tools/summarize_pmc.py $OUT/traffic_step sdpa_fwd sdpa_bwd_dq sdpa_bwd_dkv sdpa_bwd_delta > $OUT/traffic_step.json
python3 tools/make_pmc_json.py $OUT > $OUT/sdpa_pmc.json
bash tools/pmc_sdpa.sh $tag > /dev/null 2>&1; mkdir -p $OUT/sq; cp gpurun_out/pmc_sdpa_$tag/summary.json $OUT/sq/summary.json
python3 bench.py > $OUT/bench_stdout.log 2>&1
rm -rf $OUT/stats $OUT/traffic $OUT/traffic_step gpurun_out/pmc_sdpa_$tag
tail -c 600 $OUT/bench_stdout.log
