#!/usr/bin/env python
"""Reads gpurun_out/gemm_sweep_b<B>/ (tools/gemm_sweep.sh): per GEMM shape the picker's configuration and time against the best
forced one.   python tools/gemm_sweep_table.py [B]"""
import collections
import glob
import os
import re
import sys

B = sys.argv[1] if len(sys.argv) > 1 else "64"
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm_sweep_b" + B)
pat = re.compile(r"GEMM (\w+) rows=(\d+) cols=(\d+) depth=(\d+) tile=(\d+x\d+) ks=(\d+)\s+([\d.]+) us")


def read(path):
    out = collections.defaultdict(list)
    for line in open(path):
        m = pat.match(line)
        if m:
            out[(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))].append((m.group(5), int(m.group(6)), float(m.group(7))))
    return out


picked = read(os.path.join(D, "picked.txt"))
best = {}
for f in glob.glob(os.path.join(D, "force_*.txt")):
    for shape, rows in read(f).items():
        for tile, ks, us in rows:
            key = (tile, ks)
            best.setdefault(shape, {}).setdefault(key, []).append(us)
tot_p = tot_b = 0.0
for shape, rows in sorted(picked.items(), key=lambda kv: -min(r[2] for r in kv[1]) * len(kv[1])):
    tp = min(r[2] for r in rows)
    cand = sorted(((min(v), k) for k, v in best.get(shape, {}).items()))
    tb, kb = cand[0] if cand else (tp, None)
    n = len(rows)
    tot_p += n * tp
    tot_b += n * min(tp, tb)
    print("%-2s %5d x %4d x %4d  x%d  picked %-8s ks=%-2d %7.1f us | best %-8s ks=%-2d %7.1f us  %+6.1f" % (
        shape[0], shape[1], shape[2], shape[3], n, rows[0][0], rows[0][1], tp, kb[0] if kb else "-", kb[1] if kb else 0, tb, (tb - tp) * n))
print("sum picked %.1f us, sum best %.1f us" % (tot_p, tot_b))
