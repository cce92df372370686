# round-6 profiles: eval chain (kernel trace + 4 PMC passes + traffic.json), training step (kernel trace, 4 PMC passes, traffic_train.json), B=1 step stats
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r06 > gpurun_out/prof_round_r06.log 2>&1
bash tools/prof_train.sh r06t > gpurun_out/prof_train_r06.log 2>&1
bash tools/prof_train_pmc.sh r06tp r06 > gpurun_out/prof_train_pmc_r06.log 2>&1
bash tools/prof_train.sh r06t1 --batch-per-gpu 1 > gpurun_out/prof_train_b1_r06.log 2>&1
ls gpurun_out/prof_r06 gpurun_out/prof_r06t gpurun_out/prof_r06tp gpurun_out/prof_r06t1
