cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/t_train6.log 2>&1
echo "train tests rc=$?" >> gpurun_out/t_train6.log
for b in 64 64; do
timeout 600 python bench.py --mode train --batch-per-gpu $b --steps 10 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows > gpurun_out/b_m.json 2> gpurun_out/b_m.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_m.json') if l.startswith('{')][-1]);print('train B=$b', r['ms_per_step'])"
done
tail -3 gpurun_out/t_train6.log
