# training leg with the RCCL path forced (one rank), phase timing of the training recurrences, xmap A/B, B=1 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --mode train --batch-per-gpu 64 --steps 8 --warmup 2 --force-dist > gpurun_out/b_train_fd.json 2> gpurun_out/b_train_fd.err
echo "bench rc=$?" >> gpurun_out/b_train_fd.err
for x in 0 1 2 3; do
  timeout 300 python bench.py --mode train --batch-per-gpu 64 --steps 8 --warmup 2 --headline-only --no-cpu-baseline --no-other-flows --lib-option train_xmap=$x > gpurun_out/b_xmap$x.json 2> gpurun_out/b_xmap$x.err
done
timeout 300 python tools/train_phase_timing.py 64 80 > gpurun_out/phase64.log 2>&1
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "three_rows or vs_cpu_checker or forms_agree" > gpurun_out/t_ll.log 2>&1
echo "ll tests rc=$?" >> gpurun_out/t_ll.log
for w in 0 1; do
  timeout 300 python bench.py --mode train --batch-per-gpu 1 --steps 20 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows --lib-option ll_wide_rows=$w > gpurun_out/b_b1_wide$w.json 2> gpurun_out/b_b1_wide$w.err
done
tail -n 3 gpurun_out/b_train_fd.err gpurun_out/t_ll.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_xmap*.json'))+sorted(glob.glob('gpurun_out/b_b1_wide*.json'))+['gpurun_out/b_train_fd.json']:
    try:
        r=json.load(open(f)); print(f, r['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/phase64.log | tail -20
