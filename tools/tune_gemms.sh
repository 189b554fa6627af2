#!/bin/bash
# Regenerates halva_amd/tuned/gfx950_llava7b_T2048_tunableop.csv: one bench step with PyTorch TunableOp measuring every GEMM shape
# of the headline workload (about 16 minutes on one MI355X).  Run on the GPU box; copy the result from gpurun_out/ afterwards.
mkdir -p gpurun_out
export HALVA_GEMM_TABLE=$PWD/gpurun_out/tunableop_new.csv HALVA_GEMM_TUNE=1
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10
python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@"
