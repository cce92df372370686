# round-4 profiles: eval chain (kernel trace + 4 PMC passes + traffic.json), training step (kernel trace, 4 PMC passes, traffic_train.json), B=1 step stats
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r04 > gpurun_out/prof_round_r04.log 2>&1
bash tools/prof_train.sh r04t > gpurun_out/prof_train_r04.log 2>&1
bash tools/prof_train_pmc.sh r04tp > gpurun_out/prof_train_pmc_r04.log 2>&1
bash tools/prof_gaps_train.sh 1 > gpurun_out/gaps_b1.log 2>&1
ls gpurun_out/prof_r04 gpurun_out/prof_r04t gpurun_out/prof_r04tp gpurun_out/prof_gaps_train
