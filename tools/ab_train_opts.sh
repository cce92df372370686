# usage (on the GPU box): bash tools/ab_train_opts.sh [B] "<opt=v opt=v>" "<opt=v ...>" ...  -- stage-4 step per option SET
cd $GRAFT_REPO_ROOT
B=$1
shift
for set in "$@"; do
args=""
for kv in $set; do case $kv in step.*) args="$args --step-option ${kv#step.}";; *) args="$args --lib-option $kv";; esac; done
python bench.py --mode train --batch-per-gpu $B --steps 8 --warmup 2 --no-cpu-baseline --headline-only --no-other-flows $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train_step', d)
k=t.get('roofline',{}).get('kernels',{})
print('[$set] B', $B, 'ms/step', round(t['ms_per_step'],3), ' '.join('%s %.2f' % (n.split('_')[0], v.get('kernel_ms_per_step', 0)) for n, v in k.items()))"
done
