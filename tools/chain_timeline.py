#!/usr/bin/env python
"""Timeline of ONE eval chain from a rocprofv3 kernel trace (the last 8 k_gru_steps_v6 launches and what sits between them).
python tools/chain_timeline.py <kt_results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_gru_steps_v6" in r[0]]
lo = idx[-8] - 2
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0 = rows[lo][1]
prev_end = t0
for r in rows[lo:idx[-1] + 3]:
    name = r[0].split("(")[0].replace("void ", "")[:40]
    out.write("%9.1f %8.1f  gap %6.1f  %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, name))
    prev_end = r[2]
