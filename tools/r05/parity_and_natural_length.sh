#!/bin/bash
# round 5 GPU job: the full-width decoder-layer parity test, then the recipe's natural-length regime (responses of ~128 tokens, 8 and 16 pairs
# per GPU) as bench lines + a kernel-time / wall-time ratio from rocprofv3 --kernel-trace --stats.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_fullwidth_layer_parity_gpu.py -x -q -m gpu -s > $O/r05_fullwidth_parity.log 2>&1; tail -15 $O/r05_fullwidth_parity.log
cd /tmp && export TMPDIR=/tmp
for ppg in 8 16; do
  python3 $R/bench.py --resp-len 128 --pairs-per-gpu $ppg --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_natural_p$ppg.json 2> $O/r05_natural_p$ppg.err
  tail -c 1500 $O/r05_natural_p$ppg.json
  rm -rf /tmp/prof_nat_$ppg
  rocprofv3 --kernel-trace --stats -d /tmp/prof_nat_$ppg -o p --output-format csv -- python3 $R/bench.py --resp-len 128 --pairs-per-gpu $ppg --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r05_natural_p${ppg}_prof.json 2>/dev/null
  cp $(find /tmp/prof_nat_$ppg -name '*kernel_stats.csv' | head -1) $O/r05_natural_p${ppg}_kernel_stats.csv
done
