# usage (on the GPU box): bash tools/prof_gaps_eval.sh  -- idle gaps between the kernels of the eval chain
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_gaps_eval
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-paths --headline-only --no-train-leg > $D/kt.log 2>&1
DB=$(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1)
python - $DB <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# last 3 chains: print the sequence of (gap before, name, duration)
tail = rows[-60:]
prev = None
for n, s, e in tail:
    print("%8.1f gap  %8.1f us  %s" % ((s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, n.split("(")[0][:50]))
    prev = e
PY
python $R/tools/trace_gaps.py $DB 300 | head -12
rm -f $D/*.db $D/*/*.db
