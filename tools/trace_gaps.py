#!/usr/bin/env python3
"""tools/trace_gaps.py KERNEL_TRACE.csv SUBSTRING [PERIOD] -- durations of one kernel's dispatches and the GAPS between consecutive ones
(end of dispatch i -> start of dispatch i + 1, on the GPU's clock), from a rocprofv3 --kernel-trace CSV; with PERIOD the gaps are also
grouped by position inside a period (e.g. 16 = the mix queue's batch).  Everything else that ran in a gap is listed by name."""
import collections
import csv
import sys

path, sub = sys.argv[1], sys.argv[2]
period = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))), key=lambda t: t[0])
mine = [(s, e) for s, e, n in rows if sub in n]
mine = mine[len(mine) // 3:]  # (the first third: warm-up, clock ramp)
dur = [(e - s) / 1e3 for s, e in mine]
gap = [(mine[i + 1][0] - mine[i][1]) / 1e3 for i in range(len(mine) - 1)]
sts = [(mine[i + 1][0] - mine[i][0]) / 1e3 for i in range(len(mine) - 1)]
gs = sorted(gap)
print("%s: %d dispatches; duration avg %.2f us; start-to-start avg %.2f us; gap avg %.2f us, median %.2f, p90 %.2f, max %.2f"
      % (sub, len(mine), sum(dur) / len(dur), sum(sts) / len(sts), sum(gap) / len(gap), gs[len(gs) // 2], gs[int(len(gs) * 0.9)], gs[-1]))
if period:
    by = collections.defaultdict(list)
    # align the period on the largest gaps
    big = max(range(period), key=lambda ph: sum(gap[i] for i in range(ph, len(gap), period)))
    for i, g in enumerate(gap):
        by[(i - big) % period].append(g)
    print("gap by position in a period of %d (position 0 = the largest): " % period + "  ".join("%d: %.2f" % (k, sum(v) / len(v)) for k, v in sorted(by.items())))
others = collections.Counter()
for i in range(len(mine) - 1):
    for s, e, n in rows:
        if sub not in n and s >= mine[i][0] and s < mine[i + 1][0]:
            others[n.split("(")[0][-60:]] += 1
for n, c in others.most_common(6):
    print("  between them: %6d x %s" % (c, n))
