# usage (on the GPU box): bash tools/prof_train_pmc.sh <tag> [profiles prefix, e.g. r06]   -> gpurun_out/prof_<tag>/{pmc_*_summary.md, traffic_train.json} for the stage-4 training step (B=64)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4tp}
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
ARGS="--mode train --batch-per-gpu 64 --steps 2 --warmup 1 --no-cpu-baseline --headline-only --no-other-flows"
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --pmc $c --kernel-trace -d $D -o pmc_$c -- python $R/bench.py $ARGS > $D/pmc_$c.log 2>&1
  cd $R
  python tools/rocprof_summary.py $(ls $D/*pmc_${c}_results.db $D/*/pmc_${c}_results.db 2>/dev/null | head -1) $D/pmc_${c}_summary.md "rocprofv3 --pmc $c --kernel-trace on bench.py $ARGS" > /dev/null
done
python tools/traffic_train_from_pmc.py $(ls $D/*pmc_FETCH_SIZE_results.db $D/*/pmc_FETCH_SIZE_results.db 2>/dev/null | head -1) $(ls $D/*pmc_WRITE_SIZE_results.db $D/*/pmc_WRITE_SIZE_results.db 2>/dev/null | head -1) $D/traffic_train.json ${2:-rNN} > /dev/null
rm -f $D/*.db $D/*/*.db
ls $D
