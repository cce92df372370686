# usage (GPU box): bash tools/r6/stage6_backoff.sh "<v6_backoff>" ...   -- stage-6 sub-paths of bench.py per first-poll back-off of k_gru_steps_v6
cd $GRAFT_REPO_ROOT
for b in "$@"; do
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only --no-train-leg --lib-option v6_backoff=$b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sub_paths']
print('[v6_backoff=$b]', ' '.join('%s %.3f' % (k.replace('stage6_','')[:28], v.get('ms', v.get('ms_per_pair', v.get('ms_per_call', 0)))) for k, v in s.items() if k != 'batch_sweep'))"
done
