cd $GRAFT_REPO_ROOT
for o in 0 1 0 1; do
for b in 32; do
timeout 600 python bench.py --mode train --batch-per-gpu $b --steps 10 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows --lib-option max_rt=$o > gpurun_out/b_rt.json 2> gpurun_out/b_rt.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_rt.json') if l.startswith('{')][-1]);print('train B=$b max_rt=$o', r['ms_per_step'])"
done; done
