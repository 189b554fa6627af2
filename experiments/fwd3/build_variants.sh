#!/bin/bash
# round 4: timing variants of the plain step (MODE 2): what the row sums / the running maximum / the issue-unit cap cost   (run from this directory)
set -e
rm -f step_bench_*
build() {  # name, env...
  name=$1; shift
  env "$@" python3 gen_fwd_step.py plain > /dev/null
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o step_bench_$name step_bench.hip
}
build base
build nom3 FWD_NO_M3=1
build nom3_add FWD_NO_M3=1 FWD_LSUM=add || true
build nom3_add_c6 FWD_NO_M3=1 FWD_LSUM=add FWD_CAP=6
build nom3_none FWD_NO_M3=1 FWD_LSUM=none
build add_c6 FWD_LSUM=add FWD_CAP=6
build base_c6 FWD_CAP=6
build nom3_none_c4 FWD_NO_M3=1 FWD_LSUM=none FWD_CAP=4
python3 gen_fwd_step.py > /dev/null
