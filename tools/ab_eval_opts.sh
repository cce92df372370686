# usage (on the GPU box): bash tools/ab_eval_opts.sh "<opt=v ...>" "<opt=v ...>" ...  -- eval chain per option SET, per launch geometry
cd $GRAFT_REPO_ROOT
for set in "$@"; do
args=""
for kv in $set; do args="$args --lib-option $kv"; done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pi=d['roofline']['per_instantiation']
print('[$set] chain ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), ' '.join('%s %.1f us (%.3f)' % (k.split('_')[0] + ('2' if 'stacked' in k else ''), 1e3*v['avg_launch_ms'], v['frac']) for k, v in pi.items()))"
done
