#!/usr/bin/env python3
"""tools/trace_timeline.py KERNEL_TRACE.csv ANCHOR [N] -- the dispatches of the LAST N periods (default 1) of a rocprofv3 --kernel-trace
CSV as a timeline: a period starts at a dispatch whose name contains ANCHOR and follows a gap of the same name of more than
half the period.  Per dispatch: start and end relative to the period's first start (us), duration, queue, name."""
import csv
import sys

path, anchor = sys.argv[1], sys.argv[2]
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:], r.get("Queue_Id", "?"))
               for r in csv.DictReader(open(path))), key=lambda t: t[0])
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
# period starts: anchors that follow a non-anchor dispatch
starts = [i for k, i in enumerate(idx) if k == 0 or any(anchor not in rows[j][2] for j in range(idx[k - 1] + 1, i))]
# keep those that start a burst (the previous anchor is more than 30 % of the median start-to-start away)
for p in range(max(0, len(starts) - 1 - nper), len(starts) - 1):
    a, b = starts[p], starts[p + 1]
    t0 = rows[a][0]
    print("period %d: %d dispatches, %.1f us start to next start" % (p, b - a, (rows[b][0] - t0) / 1e3))
    last_end = {}
    for s, e, n, q in rows[a:b]:
        print("  %9.1f .. %9.1f  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
