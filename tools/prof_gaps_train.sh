# usage (on the GPU box): bash tools/prof_gaps_train.sh [B]  -- idle gaps between the kernels of the stage-4 training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-64}
D=$R/gpurun_out/prof_gaps_train
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --mode train --batch-per-gpu $B --steps 4 --warmup 2 --no-cpu-baseline --headline-only --no-other-flows > $D/kt.log 2>&1
DB=$(ls $D/*kt_results.db $D/*/kt_results.db 2>/dev/null | head -1)
python - $DB <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id from kernels order by start")) if False else list(db.execute("select name, start, end from kernels order by start"))
# find the last k_adam: the last step is between the two last k_adam launches
ad = [i for i, r in enumerate(rows) if r[0].startswith("k_adam_counted")]
lo, hi = ad[-2] + 1, ad[-1] + 1
step = rows[lo:hi]
print("last step: %d kernels, span %.3f ms" % (len(step), (step[-1][2] - step[0][1]) / 1e6))
# union busy
busy, cs, ce = 0, step[0][1], step[0][2]
for n, s, e in step[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("busy (union) %.3f ms, idle %.3f ms" % (busy / 1e6, (step[-1][2] - step[0][1] - busy) / 1e6))
PY
python $R/tools/trace_gaps.py $DB 400 | head -24
python $R/tools/rocprof_summary.py $DB $D/kt_summary_b$B.md "rocprofv3 --kernel-trace --stats on bench.py --mode train --batch-per-gpu $B --steps 4 --warmup 2 --no-cpu-baseline --headline-only" > /dev/null
rm -f $D/*.db $D/*/*.db
