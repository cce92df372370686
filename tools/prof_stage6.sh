# usage (on the GPU box): bash tools/prof_stage6.sh -- rocprofv3 kernel trace of the stage-6 forms -> gpurun_out/prof_stage6/kt_summary.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_stage6
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/tools/prof_stage6.py > $D/kt.log 2>&1
DB=$(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1)
python $R/tools/rocprof_summary.py $DB $D/kt_summary.md "rocprofv3 --kernel-trace --stats on tools/prof_stage6.py (3 x: pair, pair as a wavefront of 224-frame windows, list of 4 single pairs, ten pairs per call, list of 3 ten-pair calls; 637 / 660-frame utterances, 300-draw means)" > /dev/null
rm -f $D/*.db $D/*/*.db
head -30 $D/kt_summary.md
