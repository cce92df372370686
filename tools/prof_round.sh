cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2 -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-paths --headline-only > $R/gpurun_out/prof_r2/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/prof_r2/*kt_results.db gpurun_out/prof_r2/*/kt_results.db 2>/dev/null | head -1) gpurun_out/prof_r2/kt_summary.md "round 2: rocprofv3 --kernel-trace --stats on bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-paths --headline-only (exact-operand kernel k_gru_steps_v6)" > /dev/null
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_r2 -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sub-paths --headline-only > $R/gpurun_out/prof_r2/pmc_$c.log 2>&1
  cd $R
  python tools/rocprof_summary.py $(ls gpurun_out/prof_r2/*pmc_${c}_results.db gpurun_out/prof_r2/*/pmc_${c}_results.db 2>/dev/null | head -1) gpurun_out/prof_r2/pmc_${c}_summary.md "round 2: rocprofv3 --pmc $c on bench.py --steps 2 --warmup 1 --headline-only" > /dev/null
done
ls -la gpurun_out/prof_r2 | head -30
rm -f gpurun_out/prof_r2/*.db gpurun_out/prof_r2/*/*.db
