#!/bin/bash
# Everything profiles/ holds for one round, from the CURRENT build (run on the GPU box: HALVA_COMMIT=<sha> gpurun -- bash tools/collect_round_profiles.sh r03):
#   <tag>_bench_line.json / _bench_stdout.log        the default `python bench.py` line, with the in-kernel clock trace (HALVA_BENCH_CLOCK_TRACE)
#   <tag>_clock_trace_bench.json                       its samples
#   <tag>_step_kernel_stats.csv / _step_summary.md     ONE steady-state step: (rocprofv3 --stats of --steps 3) - (--steps 1), per kernel (tools/steady_state_stats.py)
#   <tag>_sdpa_pmc.json                                FETCH_SIZE / WRITE_SIZE of the SDPA kernels (microbench shape + in the step), tools/make_pmc_json.py
#   <tag>_sdpa_all_pmc.json                            SQ counters of the SDPA kernels (tools/pmc_sdpa.sh); <tag>_sdpa_tcc_in_step.json: its TCC hit / miss / fabric-read pass over one bench step
#   <tag>_rowops_pmc.json                              HBM TB/s of the row / loss kernels from counters (tools/pmc_rowops.sh)
#   <tag>_clock_under_load.json                        shader clock per kernel kind (tools/clock_under_load.py)
tag=${1:-r03}; R=$PWD; OUT=$R/gpurun_out/$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1 3; do
  rocprofv3 --kernel-trace --stats -d $OUT/stats$n -o s --output-format csv -- python3 $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_profiled_steps$n.log 2>&1
  cp $(find $OUT/stats$n -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_steps$n.csv
done
cd $R
python3 tools/steady_state_stats.py $OUT/kernel_stats_steps1.csv $OUT/kernel_stats_steps3.csv 1 3 $OUT/${tag}_step_kernel_stats.csv $OUT/${tag}_step_summary.md \
  "One steady-state bench step (7B, T=2048, 16 pairs, 1x MI355X), commit ${HALVA_COMMIT:-unrecorded}" > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $OUT/traffic/$c -o t --output-format csv -- python3 $R/tools/bench_sdpa.py > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $OUT/traffic_step/$c -o t --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
done
python3 tools/summarize_pmc.py $OUT/traffic sdpa_fwd sdpa_bwd_dq sdpa_bwd_dkv sdpa_bwd_delta > $OUT/traffic.json
python3 tools/summarize_pmc.py $OUT/traffic_step sdpa_fwd sdpa_bwd_dq sdpa_bwd_dkv sdpa_bwd_delta > $OUT/traffic_step.json
python3 tools/make_pmc_json.py $OUT > $OUT/${tag}_sdpa_pmc.json
bash tools/pmc_sdpa.sh $tag > /dev/null 2>&1; cp gpurun_out/pmc_sdpa_$tag/summary.json $OUT/${tag}_sdpa_all_pmc.json; cp gpurun_out/pmc_sdpa_$tag/summary_step.json $OUT/${tag}_sdpa_tcc_in_step.json
bash tools/pmc_rowops.sh $tag > $OUT/rowops.log 2>&1; cp gpurun_out/${tag}_rowops_pmc.json $OUT/
python3 tools/clock_under_load.py $OUT/${tag}_clock_under_load.json > $OUT/clock_under_load.log 2>&1
# the bench line last: it quotes the traffic record collected above when that has been copied to profiles/ (the copy below makes it so on the box)
cp $OUT/${tag}_sdpa_pmc.json profiles/${tag}_sdpa_pmc.json
HALVA_BENCH_CLOCK_TRACE=$OUT/${tag}_clock_trace_bench.json python3 bench.py > $OUT/${tag}_bench_stdout.log 2>&1
grep '^{' $OUT/${tag}_bench_stdout.log | tail -1 > $OUT/${tag}_bench_line.json
rm -rf $OUT/stats1 $OUT/stats3 $OUT/traffic $OUT/traffic_step gpurun_out/pmc_sdpa_$tag gpurun_out/pmc_rowops_$tag
ls -la $OUT; cat $OUT/${tag}_step_summary.md; tail -c 1200 $OUT/${tag}_bench_stdout.log
