# usage (on the GPU box): bash tools/prof_train.sh <tag> [extra bench.py args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3t}; shift
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --mode train --batch-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-other-flows "$@" > $D/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1) $D/kt_summary.md "rocprofv3 --kernel-trace --stats on bench.py --mode train --batch-per-gpu 64 --steps 3 --warmup 1 $* (4 steps in the trace)" > /dev/null
rm -f $D/*.db $D/*/*.db
tail -3 $D/kt.log
