#!/bin/bash
cd $GRAFT_REPO_ROOT/experiments/issue_cost; ./dma_cost | tee $GRAFT_REPO_ROOT/gpurun_out/r04_dma_cost.log
