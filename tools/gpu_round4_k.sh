cd $GRAFT_REPO_ROOT
timeout 1200 python -c "
import sys; sys.path[:0]=['.','cyclevae-vc_amd','tests']
import gru_vae; gru_vae._lib().set_option('train_bp16',1)
import pytest; sys.exit(pytest.main(['tests/test_gpu_train.py','-x','-q','-m','gpu','-k','vs_cpu_checker or forms_agree or autograd_vs_reference or recorded']))" > gpurun_out/t_bp16.log 2>&1
echo "bp16 tests rc=$?" >> gpurun_out/t_bp16.log
for o in 0 1 0 1; do
for b in 8 16; do
timeout 600 python bench.py --mode train --batch-per-gpu $b --steps 10 --warmup 3 --headline-only --no-cpu-baseline --no-other-flows --lib-option train_bp16=$o > gpurun_out/b_bp16_$o_$b.json 2> gpurun_out/b_bp16.err
python -c "
import json;r=json.loads([l for l in open('gpurun_out/b_bp16_$o_$b.json') if l.startswith('{')][-1]);print('train B=$b bp16=$o', r['ms_per_step'])"
done; done
tail -3 gpurun_out/t_bp16.log
