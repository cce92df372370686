# usage (on the GPU box): bash tools/ab_option.sh <option> "<values>"  -- eval chain per value of a library option
cd $GRAFT_REPO_ROOT
for v in $2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg --lib-option $1=$v | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', $v, 'eval ms', round(d['ms_per_step'],4), 'kernel us', round(1e3*d['roofline']['avg_launch_ms'],1), 'whole', round(d['whole_job']['frac_of_f32_mfma_peak'],4))"
done
