#!/bin/bash
# SQ counters of the SDPA kernels at the 7B per-layer shape (run on the GPU box: `gpurun -- bash tools/pmc_sdpa.sh [tag]`).
# Passes of <= 8 SQ counters each (PMC runs carry --kernel-trace only); writes gpurun_out/pmc_sdpa_<tag>/{a,b,c}/ and summary.json.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sdpa_${1:-run}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $OUT/a -o a --output-format csv -- python3 tools/bench_sdpa.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/b -o b --output-format csv -- python3 tools/bench_sdpa.py > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM -d $OUT/c -o c --output-format csv -- python3 tools/bench_sdpa.py > $OUT/c.log 2>&1
# L2 (TCC) hits / misses and the read requests that leave it, summed over the channels (round 4: how much of the tile traffic the XCD's L2 serves)
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/d -o d --output-format csv -- python3 tools/bench_sdpa.py > $OUT/d.log 2>&1
python3 tools/summarize_pmc.py $OUT > $OUT/summary.json
# the same TCC pass over ONE bench step (round 5: the hit rate and the fabric reads at the step's own launch shapes, not only at the micro shape)
mkdir -p $OUT/step
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/step/d -o d --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/step_d.log 2>&1
python3 tools/summarize_pmc.py $OUT/step > $OUT/summary_step.json
tail -2 $OUT/c.log
