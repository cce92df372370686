cd $GRAFT_REPO_ROOT
for T in 20 40 80 160; do
python bench.py --frames $T --steps 30 --warmup 5 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg --profile-every 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pi=d['roofline']['per_instantiation']
print('T', $T, 'chain ms', round(d['ms_per_step'],4), ' '.join('%s %.1f' % (k.split('_')[0] + ('2' if 'stacked' in k else ''), 1e3*v['avg_launch_ms']) for k, v in pi.items()))"
done
