#!/bin/bash
cd $GRAFT_REPO_ROOT/experiments/dkv3
for b in step_bench_*; do [ -x $b ] || continue; echo "== $b"; timeout 120 ./$b 2048 1024 2>&1 | grep -i "mode 2" | head -3; done
