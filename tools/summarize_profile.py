#!/usr/bin/env python3
"""tools/summarize_profile.py <tag> -- condense gpurun_out/prof_<tag>/ into profiles/<tag>_*.

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats summary, verbatim),
profiles/<tag>_summary.md and profiles/pmc_traffic.json (HBM bytes per launch of the dominant
kernel, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE under-reports a wide coalesced read stream by 2x, so it is doubled)."""
import csv
import json
import os
import shutil
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel_key = sys.argv[2] if len(sys.argv) > 2 else "osc_kernel"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "kt", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))

rows = list(csv.DictReader(open(os.path.join(src, "kt", "bench_kernel_trace.csv"))))
k = [r for r in rows if kernel_key in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in k]
dur_timed = dur[-500:]
s = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k)[-500:]
gaps = [(s[i + 1][0] - s[i][1]) / 1e3 for i in range(len(s) - 1)]


def pmc(name, sub):
    rr = [r for r in csv.DictReader(open(os.path.join(src, sub, "bench_counter_collection.csv")))
          if kernel_key in r["Kernel_Name"] and r["Counter_Name"] == name]
    return statistics.mean(float(r["Counter_Value"]) for r in rr), len(rr)


w_kib, nw = pmc("WRITE_SIZE", "pmc_w")
r_kib, nr = pmc("FETCH_SIZE", "pmc_r")
write_b = w_kib * 1024
read_b = r_kib * 1024 * 2  # gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section)
traffic = write_b + read_b
bench = {}
try:
    bench = json.loads(open(os.path.join(src, "bench_unprofiled.json")).read().strip().splitlines()[-1])
except Exception:
    pass
json.dump({"k1_hbm_bytes_per_launch": round(traffic), "write_bytes": round(write_b),
           "read_bytes_corrected": round(read_b), "source": "profiles/%s_summary.md" % tag,
           "kernel": k[0]["Kernel_Name"] if k else None},
          open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
    f.write("# rocprofv3 summary `%s` (MI355X, bench.py --steps 500 --warmup 50)\n\n" % tag)
    f.write("Dominant kernel: `%s`\n\n" % (k[0]["Kernel_Name"] if k else "?"))
    f.write("| quantity | value |\n|---|---|\n")
    f.write("| launches in trace | %d |\n" % len(dur))
    f.write("| avg duration, last 500 launches (kernel-trace) | %.2f us |\n" % statistics.mean(dur_timed))
    f.write("| median / min / max | %.2f / %.2f / %.2f us |\n" % (statistics.median(dur_timed), min(dur_timed), max(dur_timed)))
    f.write("| median gap between consecutive launches | %.2f us |\n" % statistics.median(gaps))
    f.write("| VGPR / SGPR / LDS / workgroup / grid | %s / %s / %s / %s / %s |\n" % (
        k[0]["VGPR_Count"], k[0]["SGPR_Count"], k[0]["LDS_Block_Size"], k[0]["Workgroup_Size_X"], k[0]["Grid_Size_X"]))
    f.write("| WRITE_SIZE per launch (%d samples) | %.0f KiB = %.1f MB |\n" % (nw, w_kib, write_b / 1e6))
    f.write("| FETCH_SIZE per launch (%d samples) | %.0f KiB raw, x2 gfx950 correction = %.2f MB |\n" % (nr, r_kib, read_b / 1e6))
    f.write("| HBM traffic per launch (PMC) | %.1f MB |\n" % (traffic / 1e6))
    if bench:
        rf = bench["roofline"]
        f.write("| algorithmic bytes per launch | %.1f MB |\n" % (rf["algorithmic_bytes_per_launch"] / 1e6))
        f.write("| bench.py (un-profiled) kernel_ms from HIP events | %.2f us |\n" % (rf["kernel_ms"] * 1e3))
        f.write("| bench.py achieved / peak / frac | %.0f GB/s / %.0f GB/s / %.3f |\n" % (rf["achieved"], rf["peak"], rf["frac"]))
        f.write("| bench.py value | %.0f %s |\n" % (bench["value"], bench["unit"]))
        f.write("\nUn-profiled bench line:\n\n```json\n%s\n```\n" % json.dumps(bench))
    f.write("\nrocprofv3 --stats table: `profiles/%s_kernel_stats.csv`.  Profiled passes run at lower "
            "clocks than un-profiled ones (MI355X_MICROARCH.md, DVFS), so the trace duration is an upper "
            "bound on the HIP-event duration.\n" % tag)
print(open(os.path.join(dst, tag + "_summary.md")).read())
