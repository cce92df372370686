# (1) x3h forward recurrence with counted operand waits: parity tests, phase timing, training bench; (2) word-exchange back-off sweep continued
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_cabi_exports.py -x -q -m gpu > gpurun_out/t_train2.log 2>&1
echo "train tests rc=$?" >> gpurun_out/t_train2.log
timeout 300 python tools/train_phase_timing.py 64 80 > gpurun_out/phase64_b.log 2>&1
timeout 600 python bench.py --mode train --batch-per-gpu 64 --steps 8 --warmup 2 --headline-only --no-cpu-baseline --no-other-flows > gpurun_out/b_train_x3h.json 2> gpurun_out/b_train_x3h.err
rm -f gpurun_out/ll_sweep2.log
for b in 20 22 24 26 28 32; do
  echo "== ll_backoff=$b" >> gpurun_out/ll_sweep2.log
  timeout 300 python tools/b1_timing.py ll_backoff=$b 2>&1 | grep -v amdgpu.ids | head -3 >> gpurun_out/ll_sweep2.log
done
for b in 18 20; do
timeout 200 python tools/step_timing.py 1 637 ll_backoff=$b 2>&1 | grep -v amdgpu.ids >> gpurun_out/ll_sweep2.log
done
for b in 18 22 24 28; do
  timeout 300 python bench.py --mode train --batch-per-gpu 1 --steps 20 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows --lib-option ll_backoff=$b > gpurun_out/b_b1_bo$b.json 2> gpurun_out/b_b1_bo$b.err
  python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_b1_bo$b.json') if l.startswith('{')][-1]);print('train B=1 ll_backoff=$b', r['ms_per_step'])" >> gpurun_out/ll_sweep2.log
done
tail -3 gpurun_out/t_train2.log; cat gpurun_out/phase64_b.log | tail -3; python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_train_x3h.json') if l.startswith('{')][-1]);print('train B=64', r['ms_per_step']); print({k:(round(v['kernel_ms_per_step'],2), round(v['frac'],3)) for k,v in r['roofline']['kernels'].items()})"
cat gpurun_out/ll_sweep2.log
