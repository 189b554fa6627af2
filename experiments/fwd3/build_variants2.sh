#!/bin/bash
# round 4: what the tile requests of MODE 3 cost and where they are cheapest (row sums as v_add_f32, 6 units per gap throughout)   (run from this directory)
rm -f step_bench_*
build() { name=$1; shift; env FWD_LSUM=add FWD_CAP=6 "$@" python3 gen_fwd_step.py > /dev/null && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -w -o step_bench_$name step_bench.hip; }
build d0
build d1_m0once FWD_DMA_M0=once
build d2_noreq FWD_DMA_REQ=0
build d3_nobar FWD_DMA_BAR=0
build d4_free FWD_DMA_GAPS=62,63,0,30,31,32,1,33
build d5_spread FWD_DMA_GAPS=0,8,16,24,32,40,48,56
build d6_spread_c3 FWD_DMA_GAPS=0,8,16,24,32,40,48,56 FWD_DMA_COST=3
build d7_head_c3 FWD_DMA_COST=3
build d8_cap7 FWD_CAP=7
python3 gen_fwd_step.py > /dev/null
