#!/usr/bin/env python3
"""tools/summarize_sq.py <tag> -- per-kernel shader-core counters from gpurun_out/sq_<tag>/ (tools/pmc_sq.sh) into
profiles/<tag>_sq_counters.md (averages per launch; SQ_*_CYCLES are in units of 4 shader clocks per SIMD-wave slot)."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "sq_" + tag)
acc = collections.OrderedDict()
for cfg in sorted(os.listdir(src)):
    if not os.path.isdir(os.path.join(src, cfg)):
        continue
    for f in glob.glob(os.path.join(src, cfg, "g*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "rocclr" in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("mxg::(anonymous namespace)::", "").split("(")[0]
            acc.setdefault((cfg, name), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
        "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
out = ["# Shader-core counters per launch (rocprofv3 --pmc, MI355X, round 1)", "",
       "`tools/pmc_sq.sh %s` (three passes of <= 4 SQ counters per tool), condensed by `tools/summarize_sq.py`; values in" % tag,
       "millions per launch, averaged over the launches of the kernel.  `valu busy` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x",
       "(waves per SIMD) is not derivable without occupancy, so the ratios given are per wave-cycle: the share of a resident",
       "wave's cycles in which it was issuing a VALU instruction, and the share it spent waiting on an instruction dependency.", "",
       "| tool | kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " | VALU/wave-cycle | wait/wave-cycle | LDS conflict |",
       "|---|---|" + "---|" * (len(cols) + 3)]
for (cfg, name), d in acc.items():
    m = {c: (sum(d[c]) / len(d[c]) if d.get(c) else float("nan")) for c in cols}
    if not (m["SQ_INSTS_VALU"] > 1e5):
        continue
    row = ["%.2f" % (m[c] / 1e6) for c in cols]
    wc = m["SQ_WAVE_CYCLES"]
    out.append("| %s | `%s` | %s | %.2f | %.2f | %.2f |" % (
        cfg, name, " | ".join(row), m["SQ_ACTIVE_INST_VALU"] / wc, m["SQ_WAIT_INST_ANY"] / wc,
        (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]) if m["SQ_LDS_IDX_ACTIVE"] > 0 else 0.0))
out += ["", "Reading: K1 (`osc_kernel<8>` = sinebuf) spends 60 % of its LDS-active cycles in bank conflicts -- the 64 lanes of a",
        "wavefront look up 64 unrelated entries of the 514-entry table -- yet issues VALU work in only 37 % of its wave-cycles: the",
        "conflicts are hidden behind the store stream (the kernel is HBM-write bound).  K3 (`mix_bus_kernel`) waits on memory 42 % of",
        "the time with 5 % VALU: read-bound as intended.  The fused voice (`voice_kernel<0>`, one wavefront per SIMD) issues in half of",
        "its wave-cycles (11.5 M of 22.8 M; 18.7 M of 33.8 M before the shared gate moved from scalar loads to `v_readlane`).  K6a FFT: 21 % of LDS cycles are conflicts, VALU issue in 27-38 % of wave-cycles at 4 waves per SIMD (i.e. the VALU pipe",
        "itself is the shared bottleneck).  K8c (`granular_unit_kernel`, one launch per time slice = a quarter of the call): 41 M VALU",
        "instructions per slice (445 M per call before the flattened interior pass, 164 M now); each of the four resident waves of a",
        "SIMD issues in 28 % of its cycles (24 instructions per (grain, tile) pair) and waits on a dependency in 17 %.", ""]
open(os.path.join(ROOT, "profiles", tag + "_sq_counters.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
