# word-exchange kernels after the prefetch reordering: parity tests, then back-off sweeps (eval single utterance / stage-6 pair, B=1 training step)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -m gpu -k "three_rows or word_exchange or stage6 or small_batch or vs_cpu_checker or T1500 or utterance" > gpurun_out/t_ll2.log 2>&1
echo "ll tests rc=$?" >> gpurun_out/t_ll2.log
for b in -1 8 10 12 14 16 18 20; do
  echo "== ll_backoff=$b" >> gpurun_out/ll_sweep.log
  timeout 300 python tools/b1_timing.py ll_backoff=$b 2>&1 | grep -v amdgpu.ids >> gpurun_out/ll_sweep.log
done
for b in -1 8 12 16 20; do
  timeout 300 python bench.py --mode train --batch-per-gpu 1 --steps 20 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows --lib-option ll_backoff=$b > gpurun_out/b_b1_bo$b.json 2> gpurun_out/b_b1_bo$b.err
  python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_b1_bo$b.json') if l.startswith('{')][-1]);print('train B=1 ll_backoff=$b', r['ms_per_step'])" >> gpurun_out/ll_sweep.log
done
tail -3 gpurun_out/t_ll2.log; cat gpurun_out/ll_sweep.log
