#!/usr/bin/env python
"""profiles/traffic_train.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the training leg
(tools/prof_train_pmc.sh): fabric-side bytes per launch of the training step's kernel classes, keyed like
bench.py's train_step.roofline.kernels.

    python tools/traffic_train_from_pmc.py <FETCH_SIZE db> <WRITE_SIZE db> profiles/traffic_train.json

bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane
reads on gfx950; averaged over the launches of the class's kernels in the trace.
"""
import json
import sqlite3
import sys

CLASSES = {
    "fwd_recurrence": ("k_train_fwd_steps_x3", "k_train_fwd_steps_w3"),   # x3h<8,4> (64-row passes), w3<8,4,11> (stacked 128-row passes, round 6)
    "bwd_recurrence": ("k_train_bwd_steps_x3", "k_train_bwd_steps_w3"),   # x3<32> (64-row passes), w3<8,4,12> (128-row passes, round 6)
    "forward_and_dgrad_gemms": ("k_gemm_nt2",),
    "wgrad_gemms": ("k_gemm_tn2",),
}


def per_dispatch(db_path, counter, patterns):
    db = sqlite3.connect(db_path)
    out = []
    for pat in patterns:
        out += [(n, v) for n, _, v in db.execute(
            "select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? and name like ? "
            "group by name, dispatch_id order by dispatch_id", (counter, "%" + pat + "%"))]
    return out


def main():
    fdb, wdb, out = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "rNN"        # the round's file-name prefix under profiles/
    doc = {}
    for cls, pats in CLASSES.items():
        fetch, write = per_dispatch(fdb, "FETCH_SIZE", pats), per_dispatch(wdb, "WRITE_SIZE", pats)
        if not fetch or not write:
            continue
        f = sum(v for _, v in fetch) / len(fetch)
        w = sum(v for _, v in write) / len(write)
        doc[cls] = {
            "bytes_per_launch": (2.0 * f + w) * 1024.0, "fetch_size_kb_raw_avg": f, "write_size_kb_avg": w,
            "launches_averaged": [len(fetch), len(write)],
            "kernels": sorted(set(n.split("(")[0] for n, _ in fetch)),
            "what": "fabric-side bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from two separate rocprofv3 --pmc passes over "
                    "bench.py --mode train --batch-per-gpu 64 (tools/prof_train_pmc.sh; per-kernel counters in profiles/%s_train_pmc_*.md), " % tag +
                    "averaged over the class's launches; Infinity-Cache hits included"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
