#!/usr/bin/env python3
"""tools/summarize_configs.py <tag> -- condense gpurun_out/prof_<tag>_configs/ (tools/profile_configs.sh) into
profiles/<tag>_<tool>_kernel_stats.csv and profiles/<tag>_configs_summary.md."""
import csv
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_%s_configs" % tag)
dst = os.path.join(ROOT, "profiles")
TITLES = {
    "voice": "config 3: fused subtractive voice, 65 536 voices x 512 (tools/bench_voice.py)",
    "spectral": "config 4: maxiFFT + maxiMFCC over 1 048 576 x 1024-point frames (tools/bench_spectral.py)",
    "grains": "config 5: granular time-stretch, one GPU's share (tools/bench_grains.py)",
    "banks": "bank kernels at 65 536 voices x 512 (tools/bench_banks.py)",
    "mix": "K1 / K1m / K3 mixdown variants (tools/bench_mix.py)",
}
out = ["# rocprofv3 --kernel-trace --stats, configs 3/4/5 + bank kernels (MI355X, round 1)", "",
       "Produced by `tools/profile_configs.sh %s` on the GPU box (one rocprofv3 pass per tool, then an un-profiled run of the" % tag,
       "same tool for the wall numbers) and condensed by `tools/summarize_configs.py`.  Per-kernel averages below are the",
       "profiler's; raw tables: `profiles/%s_<tool>_kernel_stats.csv`." % tag, ""]
for tool, title in TITLES.items():
    stats = os.path.join(src, tool, "k_kernel_stats.csv")
    if not os.path.exists(stats):
        continue
    shutil.copy(stats, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, tool)))
    out += ["## " + title, "", "| kernel | calls | avg (us) | min (us) | max (us) |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        name = r["Name"]
        if "rocclr" in name:
            continue
        name = name.replace("void ", "").replace("mxg::(anonymous namespace)::", "").split("(")[0]
        out.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    log = os.path.join(src, tool + ".log")
    if os.path.exists(log):
        lines = [l.rstrip() for l in open(log) if l.strip() and "amdgpu.ids" not in l]
        out += ["", "Un-profiled tool output:", "", "```"] + lines[-24:] + ["```", ""]
open(os.path.join(dst, "%s_configs_summary.md" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))
