# usage (on the GPU box): bash tools/sweep_exp.sh "<exp values>"  -- eval chain per value of the measurement switch word `exp`
cd $GRAFT_REPO_ROOT
for e in $1; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg --lib-option exp=$e | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp', $e, 'eval ms', round(d['ms_per_step'],4), 'kernel us', round(1e3*d['roofline']['avg_launch_ms'],1), 'whole', round(d['whole_job']['frac_of_f32_mfma_peak'],4))"
done
