# B=1 step anatomy, stream priority, stress train leg with its new check, full default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_gaps_train.sh 1 > gpurun_out/gaps_b1.log 2>&1
cp gpurun_out/prof_gaps_train/kt.log gpurun_out/gaps_b1_kt.log 2>/dev/null
timeout 600 python tools/train_stream_priority.py 64 80 8 > gpurun_out/prio.log 2>&1
timeout 2400 python bench.py --mode train --config stress --batch-per-gpu 64 --steps 3 --warmup 1 > gpurun_out/b_train_stress.json 2> gpurun_out/b_train_stress.err
echo "stress rc=$?" >> gpurun_out/b_train_stress.err
timeout 1500 python bench.py > gpurun_out/b_full.json 2> gpurun_out/b_full.err
echo "full rc=$?" >> gpurun_out/b_full.err
tail -n 25 gpurun_out/gaps_b1.log; cat gpurun_out/prio.log | tail -5; tail -n 4 gpurun_out/b_train_stress.err gpurun_out/b_full.err
