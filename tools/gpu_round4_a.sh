# first GPU pass of round 4: the new / changed -m gpu tests, then the training leg with the RCCL path forced (one rank)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/gpu_parity_report.txt
timeout 1700 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/t_train.log 2>&1
echo "train tests rc=$?" >> gpurun_out/t_train.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "contention or traffic" > gpurun_out/t_cont.log 2>&1
echo "contention tests rc=$?" >> gpurun_out/t_cont.log
timeout 900 python bench.py --mode train --batch-per-gpu 64 --steps 8 --warmup 2 --force-dist > gpurun_out/b_train_fd.json 2> gpurun_out/b_train_fd.err
echo "bench rc=$?" >> gpurun_out/b_train_fd.err
tail -5 gpurun_out/t_train.log gpurun_out/t_cont.log; tail -3 gpurun_out/b_train_fd.err; head -c 1500 gpurun_out/b_train_fd.json
