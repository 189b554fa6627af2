#!/bin/bash
# gpurun job: experiments/ds_residency on one MI355X -> gpurun_out/dsres_<variant>.txt
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in nt plain; do
  export HALVA_HIP_LIB=$R/halva_amd/libhalva_hip_dsres_$v.so DSRES_ORDER=/tmp/order_$v.json
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace -d /tmp/prof_$v -o p --output-format csv -- python3 $R/experiments/ds_residency/run.py > $O/dsres_$v.log 2>&1
  f=$(find /tmp/prof_$v -name '*kernel_trace.csv' | head -1)
  python3 $R/experiments/ds_residency/summarize.py $f /tmp/order_$v.json > $O/dsres_$v.txt 2>&1
  echo "== $v"; cat $O/dsres_$v.txt
done
