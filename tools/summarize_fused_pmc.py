#!/usr/bin/env python3
"""tools/summarize_fused_pmc.py PMC.json OUT.md LABEL=form ... -- the fused kernel's counters (tools/pmc_condense.py output of the
tools/gpu_runs/gpu_r05_*.sh passes) as a per-frame table, one column per form, with the derived figures the round-5 notes quote."""
import json
import sys

src, out = sys.argv[1], sys.argv[2]
cols = [a.split("=") for a in sys.argv[3:]]
d = json.load(open(src))
N = float(1 << 20)


def get(form, c):
    for k, cs in d["counters"].get(form, {}).items():
        if c in cs:
            return cs[c]["mean"] / N
    return None


names = sorted({c for f in d["counters"].values() for k in f.values() for c in k})
L = ["# `fft_mfcc_kernel`, one launch = 1 048 576 frames: counters per FRAME (rocprofv3 --pmc, separate passes; MI355X)", "",
     "`SQ_*_CYCLES`, `SQ_ACTIVE_*`, `SQ_WAIT_*` are in quad-cycles (4 shader clocks) per wavefront; `SQ_LDS_*` in LDS cycles per CU;",
     "`FETCH_SIZE` / `WRITE_SIZE` in KB as rocprofv3 reports them (FETCH_SIZE x 2 = bytes of a wide coalesced read on gfx950).", "",
     "| counter | " + " | ".join(lab for lab, _ in cols) + " |", "|---|" + "---|" * len(cols)]
for c in names:
    vals = [get(f, c) for _, f in cols]
    L.append("| %s | %s |" % (c, " | ".join("%.2f" % v if v is not None else "-" for v in vals)))
L += ["", "## Derived", "", "| figure | " + " | ".join(lab for lab, _ in cols) + " |", "|---|" + "---|" * len(cols)]


def row(name, fn):
    vals = []
    for _, f in cols:
        try:
            vals.append(fn(f))
        except Exception:
            vals.append(None)
    L.append("| %s | %s |" % (name, " | ".join(("%.3g" % v) if v is not None else "-" for v in vals)))


row("VALU instructions per frame", lambda f: get(f, "SQ_INSTS_VALU"))
row("of them packed / plain fp32 (ADD + MUL + FMA)", lambda f: get(f, "SQ_INSTS_VALU_ADD_F32") + get(f, "SQ_INSTS_VALU_MUL_F32") + get(f, "SQ_INSTS_VALU_FMA_F32"))
row("of them fp64", lambda f: get(f, "SQ_INSTS_VALU_ADD_F64") + get(f, "SQ_INSTS_VALU_MUL_F64") + get(f, "SQ_INSTS_VALU_FMA_F64"))
row("LDS instructions per frame", lambda f: get(f, "SQ_INSTS_LDS"))
row("LDS cycles per frame (IDX_ACTIVE)", lambda f: get(f, "SQ_LDS_IDX_ACTIVE"))
row("of them bank conflicts", lambda f: get(f, "SQ_LDS_BANK_CONFLICT"))
row("matrix instructions per frame", lambda f: get(f, "SQ_INSTS_MFMA"))
row("matrix pipe busy cycles per frame", lambda f: get(f, "SQ_VALU_MFMA_BUSY_CYCLES"))
row("MFMA math ops per frame (MOPS_F64 x 512)", lambda f: get(f, "SQ_INSTS_VALU_MFMA_MOPS_F64") * 512)
row("wave-cycles per frame (x4 clocks)", lambda f: get(f, "SQ_WAVE_CYCLES"))
row("share issuing (ACTIVE_INST_ANY)", lambda f: get(f, "SQ_ACTIVE_INST_ANY") / get(f, "SQ_WAVE_CYCLES"))
row("share waiting to issue (WAIT_INST_ANY)", lambda f: get(f, "SQ_WAIT_INST_ANY") / get(f, "SQ_WAVE_CYCLES"))
row("share on a wait counter (WAIT_ANY)", lambda f: get(f, "SQ_WAIT_ANY") / get(f, "SQ_WAVE_CYCLES"))
row("HBM read bytes per frame (FETCH_SIZE x 2 x 1024)", lambda f: get(f, "FETCH_SIZE") * 2 * 1024)
row("HBM write bytes per frame (WRITE_SIZE x 1024)", lambda f: get(f, "WRITE_SIZE") * 1024)
L += ["", "## Kernel-trace averages of the same commands (profiled clocks)", ""]
for k, rows in sorted(d.get("kernel_stats", {}).items()):
    for r in rows:
        L.append("* `%s`: %s calls, average %.1f us, min %.1f us" % (k, r.get("Calls"), float(r.get("AverageNs", 0)) / 1e3, float(r.get("MinNs", 0)) / 1e3))
open(out, "w").write("\n".join(L) + "\n")
print("wrote", out)
