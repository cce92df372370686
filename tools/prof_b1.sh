cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2b1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2b1 -o kt -- python $R/tools/b1_timing.py > $R/gpurun_out/prof_r2b1/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/prof_r2b1/*kt_results.db gpurun_out/prof_r2b1/*/kt_results.db 2>/dev/null | head -1) gpurun_out/prof_r2b1/kt_summary.md "round 2: rocprofv3 --kernel-trace --stats on tools/b1_timing.py (single utterance T=637 and the stage-6 pair: k_gru_steps_ll)" > /dev/null
rm -f gpurun_out/prof_r2b1/*.db gpurun_out/prof_r2b1/*/*.db
