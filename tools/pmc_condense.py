#!/usr/bin/env python3
"""tools/pmc_condense.py DIR OUT.json [kernel-substring ...] -- condense rocprofv3 --pmc outputs on the GPU box.

Walks DIR for *counter_collection.csv (one pass per sub-directory), keeps the dispatches whose kernel name contains one of the
substrings (default: every kernel that is not a runtime helper), and writes {kernel: {counter: {"mean": per-dispatch value summed
over the counter's instances, "n": dispatches}}} + the kernel-trace stats found beside them.  The raw CSVs (tens of MB with torch's
start-up kernels in them) can then be deleted before gpurun copies gpurun_out/ back (64 MiB limit)."""
import collections
import csv
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
subs = sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for root, _, files in os.walk(src):
    for fn in files:
        if not fn.endswith("counter_collection.csv"):
            continue
        for r in csv.DictReader(open(os.path.join(root, fn))):
            name = r["Kernel_Name"]
            if "rocclr" in name or (subs and not any(s in name for s in subs)):
                continue
            short = name.replace("void ", "").replace("mxg::(anonymous namespace)::", "").split("(")[0]
            key = (os.path.relpath(root, src).split(os.sep)[0], short)
            acc[key][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
res = {}
for (tag, k), cs in acc.items():
    res.setdefault(tag, {})[k] = {c: {"mean": sum(d.values()) / len(d), "n": len(d)} for c, d in cs.items()}
stats = {}
for root, _, files in os.walk(src):
    for fn in files:
        if fn.endswith("kernel_stats.csv"):
            rows = [r for r in csv.DictReader(open(os.path.join(root, fn))) if not subs or any(s in r["Name"] for s in subs)]
            stats[os.path.relpath(root, src)] = rows
json.dump({"counters": res, "kernel_stats": stats}, open(out, "w"), indent=1)
print("condensed", len(acc), "kernel/pass entries ->", out)
