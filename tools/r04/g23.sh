#!/bin/bash
cd $GRAFT_REPO_ROOT/experiments/issue_cost; ./issue_cost | tee $GRAFT_REPO_ROOT/gpurun_out/r04_issue_cost.log
