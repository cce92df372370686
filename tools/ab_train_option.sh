# usage (on the GPU box): bash tools/ab_train_option.sh <option> "<values>" [B]  -- stage-4 training step per value of a library option
cd $GRAFT_REPO_ROOT
B=${3:-64}
for v in $2; do
python bench.py --mode train --batch-per-gpu $B --steps 8 --warmup 2 --no-cpu-baseline --headline-only --no-other-flows --lib-option $1=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train_step', d)
k=t.get('roofline',{}).get('kernels',{})
print('$1', $v, 'B', $B, 'ms/step', round(t['ms_per_step'],3), ' '.join('%s %.2f' % (n, v.get('kernel_ms_per_step', 0)) for n, v in k.items()))"
done
