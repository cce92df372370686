# usage (on the GPU box): bash tools/ab_eval_exp.sh "<exp values>"  -- eval chain per value of the measurement switch `exp`, per launch geometry
cd $GRAFT_REPO_ROOT
for v in $1; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg --profile-every 1 --lib-option exp=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pi=d['roofline']['per_instantiation']
print('exp', $v, 'backoff', ($v >> 8) - 1 if $v >> 8 else 0, 'chain ms', round(d['ms_per_step'],4), ' '.join('%s %.1f us' % (k.split('_')[0] + ('2' if 'stacked' in k else ''), 1e3*v['avg_launch_ms']) for k, v in pi.items()))"
done
