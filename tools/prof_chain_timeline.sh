# usage (on the GPU box): bash tools/prof_chain_timeline.sh  -- kernel timeline of one eval chain (headline workload)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_chain
mkdir -p $D
rocprofv3 --kernel-trace -d $D -o kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg --no-kernel-events > $D/kt.log 2>&1
DB=$(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1)
python $R/tools/chain_timeline.py $DB $D/chain_timeline.txt
rm -f $D/*.db $D/*/*.db
