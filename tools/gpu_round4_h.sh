cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/t_train4.log 2>&1
echo "train tests rc=$?" >> gpurun_out/t_train4.log
for d in 1 0 1 0; do
timeout 600 python bench.py --mode train --batch-per-gpu 64 --steps 8 --warmup 2 --headline-only --no-cpu-baseline --no-other-flows --lib-option train_defer_flag=$d > gpurun_out/b_train_defer$d.json 2> gpurun_out/b_train_defer$d.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_train_defer$d.json') if l.startswith('{')][-1]);print('train B=64 defer=$d', r['ms_per_step']); print({k:(round(v['kernel_ms_per_step'],2), round(v['frac'],3)) for k,v in r['roofline']['kernels'].items()})"
done
timeout 300 python tools/train_phase_timing.py 64 80 > gpurun_out/phase64_d.log 2>&1
tail -3 gpurun_out/t_train4.log; tail -3 gpurun_out/phase64_d.log
