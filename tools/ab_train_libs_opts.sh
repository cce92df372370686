# usage (on the GPU box): bash tools/ab_train_libs_opts.sh [B] "<lib.so> opt=v opt=v" ...  -- stage-4 step per (library build, option set)
cd $GRAFT_REPO_ROOT
B=$1
shift
for set in "$@"; do
lib=${set%% *}; rest=${set#"$lib"}
args=""
for kv in $rest; do args="$args --lib-option $kv"; done
CYCLEVAE_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --mode train --batch-per-gpu $B --steps 8 --warmup 2 --no-cpu-baseline --headline-only --no-other-flows $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train_step', d)
k=t.get('roofline',{}).get('kernels',{})
print('[$set] ms/step', round(t['ms_per_step'],3), ' '.join('%s %.2f' % (n.split('_')[0], v.get('kernel_ms_per_step', 0)) for n, v in k.items()))"
done
