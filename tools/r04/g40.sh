#!/bin/bash
# steady-state kernel tables of the configs[3] / configs[4] workloads (one step = (--steps 3) - (--steps 1), per kernel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 13b vila13b; do
  for n in 1 3; do
    rocprofv3 --kernel-trace --stats -d $O/$m$n -o s --output-format csv -- python3 $R/bench.py --model $m --steps $n --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_$m$n.log 2>&1
    cp $(find $O/$m$n -name '*kernel_stats.csv' | head -1) $O/ks_$m$n.csv
    rm -rf $O/$m$n
  done
  (cd $R && python3 tools/steady_state_stats.py $O/ks_${m}1.csv $O/ks_${m}3.csv 1 3 $O/r04_step_kernel_stats_$m.csv $O/r04_step_summary_$m.md "One steady-state bench step, --model $m, 1x MI355X, commit $(cat $R/tools/r04/commit.txt)" > /dev/null)
  head -12 $O/r04_step_summary_$m.md
done
