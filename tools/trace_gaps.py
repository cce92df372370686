#!/usr/bin/env python
"""How much of a rocprofv3 kernel trace is the GPU idle?  python tools/trace_gaps.py <kt_results.db> [skip_first_ms]
Prints: wall span, union of kernel intervals, idle time, and the kernels after which the largest total idle time follows."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
rows = list(db.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1] + skip
rows = [r for r in rows if r[1] >= t0]
span = rows[-1][2] - rows[0][1]
busy, cur_s, cur_e = 0, rows[0][1], rows[0][2]
gaps = defaultdict(lambda: [0, 0])
prev_name = rows[0][0]
for name, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = gaps[(prev_name.split("(")[0][:60], name.split("(")[0][:60])]
        g[0] += s - cur_e
        g[1] += 1
        cur_s, cur_e = s, e
        prev_name = name
    else:
        if e > cur_e:
            cur_e = e
            prev_name = name
busy += cur_e - cur_s
print("span %.2f ms, busy (union) %.2f ms, idle %.2f ms (%.1f %%), sum of kernel durations %.2f ms, %d kernels" %
      (span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, sum(e - s for _, s, e in rows) / 1e6, len(rows)))
for (a, b), (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us in %4d gaps (%.1f us each)  %s -> %s" % (t / 1e3, n, t / 1e3 / n, a, b))
