cd $GRAFT_REPO_ROOT
for o in 0 1 0 1; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg --lib-option coop_launch=$o | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop', $o, 'eval ms', round(d['ms_per_step'],4), 'kernel frac', round(d['roofline']['frac'],4), 'whole', round(d['whole_job']['frac_of_f32_mfma_peak'],4))"
done
for o in 0 1; do
python bench.py --mode train --batch-per-gpu 64 --steps 8 --warmup 2 --no-cpu-baseline --headline-only --lib-option coop_launch=$o | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop', $o, 'train ms', round(d['ms_per_step'],3))"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
