cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 8 1; do
D=$R/gpurun_out/prof_b$B
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --mode train --batch-per-gpu $B --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $D/kt.log 2>&1
tail -1 $D/kt.log | cut -c1-300
DB=$(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1)
python $R/tools/trace_gaps.py $DB 200 | head -14
python $R/tools/rocprof_summary.py $DB $D/kt_summary.md "B=$B train" > /dev/null
rm -f $D/*.db $D/*/*.db
done
