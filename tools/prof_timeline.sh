# usage (on the GPU box): bash tools/prof_timeline.sh [B] [extra bench args]  -- kernel timeline of one training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-64}
shift
D=$R/gpurun_out/prof_timeline
mkdir -p $D
rocprofv3 --kernel-trace -d $D -o kt -- python $R/bench.py --mode train --batch-per-gpu $B --steps 3 --warmup 2 --no-cpu-baseline --headline-only --no-other-flows "$@" > $D/kt.log 2>&1
DB=$(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1)
python $R/tools/step_timeline.py $DB $D/timeline_b$B.txt
rm -f $D/*.db $D/*/*.db
tail -3 $D/kt.log | cut -c1-300
