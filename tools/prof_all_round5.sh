# round-5 profiles: eval chain (kernel trace + 4 PMC passes + traffic.json), training step (kernel trace, 4 PMC passes, traffic_train.json), B=1 step stats
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r05 > gpurun_out/prof_round_r05.log 2>&1
bash tools/prof_train.sh r05t > gpurun_out/prof_train_r05.log 2>&1
bash tools/prof_train_pmc.sh r05tp > gpurun_out/prof_train_pmc_r05.log 2>&1
bash tools/prof_train.sh r05t1 --batch-per-gpu 1 > gpurun_out/prof_train_b1_r05.log 2>&1
ls gpurun_out/prof_r05 gpurun_out/prof_r05t gpurun_out/prof_r05tp gpurun_out/prof_r05t1
