#!/bin/bash
# timing variants of the forward step stream (experiments/fwd3/build_variants.sh)
cd $GRAFT_REPO_ROOT/experiments/fwd3
for b in step_bench_*; do [ -x $b ] || continue; echo "== $b"; timeout 120 ./$b 256 1024 2>&1 | grep "mode [23] rep [12]"; done
