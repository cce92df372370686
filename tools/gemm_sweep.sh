# usage (on the GPU box): bash tools/gemm_sweep.sh [B]  -- every GEMM shape of an encoder + decoder train pass under every (tile, split)
# forced in turn (option gemm_force); tools/gemm_sweep_table.py turns the logs into "picked vs best" per shape
cd $GRAFT_REPO_ROOT
B=${1:-64}
D=gpurun_out/gemm_sweep_b$B
mkdir -p $D
python tools/gemm_log.py $B 80 2>&1 | grep "^GEMM" > $D/picked.txt
for t in 404 304 204 302 202 102 101; do for k in 1 2 4 8 16; do
f=$(( (t / 100) * 10000 + (t % 100) * 100 + k ))
python tools/gemm_log.py $B 80 $f 2>&1 | grep "^GEMM" > $D/force_$f.txt
done; done
ls $D | wc -l
