# usage (on the GPU box): bash tools/ab_stress_opts.sh "<opt=v opt=v>" ...  -- hu2048 / ld64 / cyc4 stage-4 step (B = 64) per option SET
cd $GRAFT_REPO_ROOT
for set in "$@"; do
args=""
for kv in $set; do args="$args --lib-option $kv"; done
python bench.py --mode train --config stress --batch-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-other-flows $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train_step', d); k=t['roofline']['kernels']
print('[$set]', round(t['ms_per_step'],2), {n.split('_')[0]: round(v['kernel_ms_per_step'],1) for n,v in k.items()})"
done
