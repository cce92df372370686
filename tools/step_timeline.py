#!/usr/bin/env python
"""Timeline of ONE stage-4 training step from a rocprofv3 kernel trace: every kernel of the last step between two k_adam launches,
with its start offset, duration and queue -- what runs beside what.  python tools/step_timeline.py <kt_results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")
rows = list(db.execute(sel))
ad = [i for i, r in enumerate(rows) if r[0].startswith("k_adam_counted")]
lo, hi = ad[-2] + 1, ad[-1] + 1
step = rows[lo:hi]
t0 = step[0][1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("# columns: %s\n# %d kernels, span %.3f ms\n" % (cols, len(step), (step[-1][2] - t0) / 1e6))
for r in step:
    name = r[0].split("(")[0].replace("void ", "")[:44]
    out.write("%9.1f %8.1f  q%-3s %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", name))
