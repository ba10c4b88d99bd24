#!/usr/bin/env python3
"""tools/line_fields.py LABEL -- reads bench.py's JSON line on stdin, prints the timing fields on one line (A/B scripts)."""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    ks = {k: round(v["ms"] * 1e3, 2) for k, v in d.get("kernels", {}).items()}
    print("%-34s ms_per_step %.4f  step_ms_gpu %s  %s %.2f us  frac %s  kernels(us) %s" % (
        label, d["ms_per_step"], d.get("step_ms_gpu"), r.get("kernel"), (r.get("kernel_ms") or 0) * 1e3, r.get("frac"), ks))
