#!/usr/bin/env python3
"""tools/summarize_pmc.py <tag> -- per-kernel HBM traffic from gpurun_out/pmc_<tag>/ (tools/pmc_configs.sh) into
profiles/<tag>_pmc_traffic.md.  WRITE_SIZE / FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports a wide
coalesced read stream by 2x and is doubled, exactly as for K1 (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
rows = collections.OrderedDict()
for cfg in sorted(os.listdir(src)):
    if not os.path.isdir(os.path.join(src, cfg)):
        continue
    for sub, cname in (("w", "WRITE_SIZE"), ("r", "FETCH_SIZE")):
        for f in glob.glob(os.path.join(src, cfg, sub, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cname or "rocclr" in r["Kernel_Name"]:
                    continue
                name = r["Kernel_Name"].replace("void ", "").replace("mxg::(anonymous namespace)::", "").split("(")[0]
                rows.setdefault((cfg, name), {"WRITE_SIZE": [], "FETCH_SIZE": []})[cname].append(float(r["Counter_Value"]))
out = ["# HBM traffic per launch from rocprofv3 --pmc (MI355X, round 1)", "",
       "`tools/pmc_configs.sh %s` (separate WRITE_SIZE and FETCH_SIZE passes per tool) condensed by" % tag,
       "`tools/summarize_pmc.py`.  KiB counters -> bytes; FETCH_SIZE doubled (gfx950 correction).  Averages over all",
       "launches of the kernel in the tool run.", "",
       "| tool | kernel | launches | written (MB) | read, corrected (MB) | total (MB) |", "|---|---|---|---|---|---|"]
for (cfg, name), d in rows.items():
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024
    r = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) * 1024 * 2
    out.append("| %s | `%s` | %d | %.1f | %.1f | %.1f |" % (cfg, name, max(len(d["WRITE_SIZE"]), len(d["FETCH_SIZE"])),
                                                         w / 1e6, r / 1e6, (w + r) / 1e6))
out += ["", "Reading (algorithmic bytes per launch in brackets): K2 filter/env 543-548 MB [537]; K4 delay 1079 MB [1074]; fused voice",
        "287 MB [280]; K1 271 MB [270]; K3 mixdown 277 MB read [268 + gains]; K6a FFT mags-only 6444 MB [4295 read + 2147 written];",
        "K5 sample players 269 MB written [268] plus 88-169 MB of gather reads that miss the caches; K8c granular 4 x 289 MB written",
        "[1156 per call, one launch per time slice] plus 4 x 277 MB of sample-buffer reads from HBM (the 35 MB buffer is re-fetched",
        "~30x: 4.9 GB of algorithmic grain reads are served mostly by L2 / Infinity Cache).  For K7a-t (MFCC, 16-B per-lane row segments) the doubled FETCH_SIZE is still only",
        "half of the 1.95 GB the kernel must read, i.e. the counter's unit depends on the request width; it is listed as measured.", ""]
open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
