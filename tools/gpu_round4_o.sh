cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/t_train7.log 2>&1
echo "train tests rc=$?" >> gpurun_out/t_train7.log
for o in 1 0 1 0; do
timeout 600 python bench.py --mode train --batch-per-gpu 64 --steps 10 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows --lib-option bwd_split_launch=$o > gpurun_out/b_o.json 2> gpurun_out/b_o.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_o.json') if l.startswith('{')][-1]);print('train B=64 split=$o', r['ms_per_step'], round(r['roofline']['kernels']['bwd_recurrence']['kernel_ms_per_step'],2))"
done
tail -3 gpurun_out/t_train7.log
