cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2t
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2t -o kt -- python $R/bench.py --mode train --batch-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r2t/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/prof_r2t/*kt_results.db gpurun_out/prof_r2t/*/kt_results.db 2>/dev/null | head -1) gpurun_out/prof_r2t/kt_summary.md "round 2: rocprofv3 --kernel-trace --stats on bench.py --mode train --batch-per-gpu 64 --steps 3 --warmup 1 (4 steps in the trace)" > /dev/null
rm -f gpurun_out/prof_r2t/*.db gpurun_out/prof_r2t/*/*.db
