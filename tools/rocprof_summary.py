#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a small text summary for profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python bench.py ...
    python tools/rocprof_summary.py gpurun_out/prof/run_results.db profiles/r01_kernel_stats.md "title"

With a --pmc run the per-dispatch counter values are summarised per kernel as well.
"""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db_path
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    lines = ["# " + title, "", "source: `%s` (rocprofv3 rocpd database, durations in microseconds)" % db_path, ""]
    lines += ["| kernel | calls | total us | avg us | min us | max us | % | vgpr | agpr | sgpr | lds |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    rows = list(cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    for r in rows:
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f | %s | %s | %s | %s |" % (
            r[0][:110], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
    try:
        # one row per (dispatch, counter instance): sum the instances of a dispatch, then average over dispatches
        pm = list(cur.execute(
            "select name, counter_name, count(*), avg(v), min(v), max(v) from (select name, counter_name, dispatch_id, "
            "sum(counter_value) as v from pmc_events group by name, counter_name, dispatch_id) group by name, counter_name "
            "order by 1, 2"))
    except sqlite3.Error as e:
        pm = []
        lines += ["", "(no counter data: %s)" % e]
    if pm:
        lines += ["", "## counters (summed over counter instances, per dispatch)", "",
                  "| kernel | counter | dispatches | avg per dispatch | min | max |", "|---|---|---|---|---|---|"]
        for r in pm:
            lines.append("| `%s` | %s | %d | %.6g | %.6g | %.6g |" % (r[0][:90], r[1], r[2], r[3], r[4], r[5]))
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
