cd $GRAFT_REPO_ROOT
for bo in -1 14 18 20 22 24 26; do
  echo "ll_backoff=$bo"
  timeout 200 python bench.py --mode train --batch-per-gpu 1 --steps 12 --warmup 3 --no-other-flows --no-kernel-events --no-cpu-baseline --lib-option ll_backoff=$bo 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('  ms_per_step', round(d.get('ms_per_step'),3))
"
done
