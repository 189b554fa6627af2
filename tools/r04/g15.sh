export HALVA_COMMIT=$(cat tools/r04/commit.txt)
bash tools/collect_round_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
tail -40 gpurun_out/r04_collect.log
