# usage (on the GPU box): bash tools/ab_train_libs.sh [B] <lib.so> <lib.so> ...  -- stage-4 step per BUILD of the library (CYCLEVAE_LIB)
cd $GRAFT_REPO_ROOT
B=$1
shift
for lib in "$@"; do
CYCLEVAE_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --mode train --batch-per-gpu $B --steps 8 --warmup 2 --no-cpu-baseline --headline-only --no-other-flows 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train_step', d)
k=t.get('roofline',{}).get('kernels',{})
print('[$lib] B', $B, 'ms/step', round(t['ms_per_step'],3), ' '.join('%s %.2f' % (n.split('_')[0], v.get('kernel_ms_per_step', 0)) for n, v in k.items()))"
done
