# usage (on the GPU box): bash tools/prof_round.sh <tag>     -> gpurun_out/prof_<tag>/{kt_summary.md, pmc_*_summary.md, traffic.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
ARGS="--no-cpu-baseline --no-sub-paths --headline-only --no-train-leg"
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --steps 5 --warmup 2 $ARGS > $D/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1) $D/kt_summary.md "rocprofv3 --kernel-trace --stats on bench.py --steps 5 --warmup 2 $ARGS (exact-operand kernel k_gru_steps_v6)" > /dev/null
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --pmc $c --kernel-trace -d $D -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 $ARGS > $D/pmc_$c.log 2>&1
  cd $R
  python tools/rocprof_summary.py $(ls $D/*pmc_${c}_results.db $D/*/pmc_${c}_results.db 2>/dev/null | head -1) $D/pmc_${c}_summary.md "rocprofv3 --pmc $c --kernel-trace on bench.py --steps 2 --warmup 1 $ARGS" > /dev/null
done
python tools/traffic_from_pmc.py $(ls $D/*pmc_FETCH_SIZE_results.db $D/*/pmc_FETCH_SIZE_results.db 2>/dev/null | head -1) $(ls $D/*pmc_WRITE_SIZE_results.db $D/*/pmc_WRITE_SIZE_results.db 2>/dev/null | head -1) $D/traffic.json k_gru_steps_v6 > /dev/null
ls -la $D | head -30
rm -f $D/*.db $D/*/*.db
