cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t_par.log 2>&1
echo "parity tests rc=$?" >> gpurun_out/t_par.log
for i in 1 2; do
timeout 900 python bench.py --no-train-leg --no-sub-paths --no-cpu-baseline --headline-only > gpurun_out/b_eval.json 2> gpurun_out/b_eval.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_eval.json') if l.startswith('{')][-1]);print(r['value'], r['ms_per_step'], r['roofline']['frac']); print({k:round(v['avg_launch_ms'],4) for k,v in r['roofline']['per_instantiation'].items()})"
done
timeout 600 python bench.py --config stress --no-cpu-baseline > gpurun_out/b_stress.json 2> gpurun_out/b_stress.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_stress.json') if l.startswith('{')][-1]);print('stress', r['value'], r['ms_per_step'], r['roofline']['frac'])"
tail -3 gpurun_out/t_par.log
