#!/bin/bash
# round 6: the batched LoRA weight-gradient launch - bitwise tests, then bench A/B (HALVA_WGRAD_BATCH=1 / 0 alternating on one box), bench at the recipe's natural lengths too
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_kernels.py -q -m gpu -k "wgrad" > $O/r06_pytest_wgrad.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_pytest_wgrad.log
timeout 900 python3 -m pytest tests/test_dpa_step_gpu.py tests/test_loss_curve_gpu.py -q -m gpu -x > $O/r06_pytest_step_wgrad.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r06_pytest_step_wgrad.log
for r in 1 2; do for b in 1 0; do
  HALVA_WGRAD_BATCH=$b python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench   batch=$b', d['value'], d['ms_per_step'])"
done; done | tee $O/r06_ab_wgrad_batch.log
for r in 1 2; do for b in 1 0; do
  HALVA_WGRAD_BATCH=$b python3 bench.py --steps 20 --warmup 3 --resp-len 128 --pairs-per-gpu 8 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('natural batch=$b', d['value'], d['ms_per_step'])"
done; done | tee -a $O/r06_ab_wgrad_batch.log
