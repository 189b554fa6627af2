#!/bin/bash
# HBM GB/s of the HBM-bound row / loss kernels from counters (run on the GPU box: gpurun -- bash tools/pmc_rowops.sh <tag>):
# three rocprofv3 passes over tools/bench_rowops.py - kernel trace + stats (durations), --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes,
# kernel-trace only beside the counters) - summarised by tools/summarize_rowops_pmc.py into gpurun_out/<tag>_rowops_pmc.json.
tag=${1:-r03}; R=$PWD; OUT=$R/gpurun_out/pmc_rowops_$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/tools/bench_rowops.py > $OUT/bench_rowops_stdout.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o t --output-format csv -- python3 $R/tools/bench_rowops.py > /dev/null 2>&1
done
cd $R
python3 tools/summarize_rowops_pmc.py $OUT > gpurun_out/${tag}_rowops_pmc.json
cat $OUT/bench_rowops_stdout.log | tail -12
python3 -c "
import json; j=json.load(open('gpurun_out/${tag}_rowops_pmc.json'))
for k,v in j['kernels'].items(): print('%-28s %8.1f us  counters %.2f TB/s  algorithmic %.2f TB/s' % (k, v['avg_us'], v['hbm_tb_s_from_counters'], v.get('algorithmic_tb_s') or 0))
"
