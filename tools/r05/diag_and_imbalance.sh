#!/bin/bash
# round 5 GPU job: (1) the SDPA kernel tests incl. the new repair test, (2) which rounding carries the long fixture's error (tools/diag_long_fixture.py),
# (3) the predicted 8-rank imbalance at natural lengths (tools/predict_imbalance.py, grouped and plain random order), (4) one steady-state step
# at natural length by kernel (8 pairs, responses of 128 tokens), (5) the step fixtures with -s (the product's own margin / gradient errors).
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "sdpa" > $O/r05_pytest_sdpa.log 2>&1; tail -3 $O/r05_pytest_sdpa.log
timeout 1200 python3 tools/diag_long_fixture.py dpa_step_d128_long > $O/r05_diag_long_fixture.log 2>&1; cat $O/r05_diag_long_fixture.log
timeout 900 python3 tools/predict_imbalance.py --steps 4 > $O/r05_imbalance_grouped.json 2> $O/r05_imbalance_grouped.err; tail -5 $O/r05_imbalance_grouped.err; tail -c 600 $O/r05_imbalance_grouped.json
timeout 900 python3 tools/predict_imbalance.py --steps 4 --no-grouping > $O/r05_imbalance_random.json 2> $O/r05_imbalance_random.err; tail -5 $O/r05_imbalance_random.err
timeout 900 python3 -m pytest tests/test_dpa_step_gpu.py -x -q -m gpu -s -k "test_step_matches_reference_golden" > $O/r05_pytest_step_s.log 2>&1; grep -E "margin err|gradient error|passed|failed" $O/r05_pytest_step_s.log | tail -40
cd /tmp && export TMPDIR=/tmp
for n in 1 3; do
  rm -rf /tmp/nat$n
  rocprofv3 --kernel-trace --stats -d /tmp/nat$n -o s --output-format csv -- python3 $R/bench.py --resp-len 128 --pairs-per-gpu 8 --steps $n --warmup 2 --no-cpu-baseline --no-roofline > $O/r05_natural_steps$n.log 2>&1
  cp $(find /tmp/nat$n -name '*kernel_stats.csv' | head -1) $O/r05_natural_kernel_stats_steps$n.csv
done
cd $R
python3 tools/steady_state_stats.py $O/r05_natural_kernel_stats_steps1.csv $O/r05_natural_kernel_stats_steps3.csv 1 3 $O/r05_natural_step_kernel_stats.csv $O/r05_natural_step_summary.md \
  "One steady-state step at the recipe's natural length (7B, responses of 128 tokens: T = 757 / 857 packed, 8 pairs, 1x MI355X)" > /dev/null
head -30 $O/r05_natural_step_summary.md
