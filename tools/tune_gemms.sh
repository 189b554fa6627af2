#!/bin/bash
# Measures the GEMM shapes of one bench workload with PyTorch TunableOp (about 16 minutes per workload on one MI355X) and writes
# gpurun_out/tunableop_<tag>.csv; merge into the shipped table with tools/merge_gemm_tables.py.  Run on the GPU box, e.g.
#   gpurun -- tools/tune_gemms.sh 7b            |  tools/tune_gemms.sh 13b --model 13b  |  tools/tune_gemms.sh vila13b --model vila13b
tag=${1:-7b}; shift
mkdir -p gpurun_out
export HALVA_GEMM_TABLE=$PWD/gpurun_out/tunableop_$tag.csv HALVA_GEMM_TUNE=1
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10
python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | cut -c1-200
wc -l $HALVA_GEMM_TABLE
