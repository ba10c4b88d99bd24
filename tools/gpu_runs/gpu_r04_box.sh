#!/bin/bash
# round 4: the default bench line once per box (called several times: box-to-box spread of the headline and of the configs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04box
mkdir -p $O
cd $R
T=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$T.json 2> $O/bench_$T.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04box/bench_$T.json") if l.startswith("{")][0])
print("box $T headline step_ms_gpu", d["step_ms_gpu"], "frac", d["roofline"]["frac"], "write ceiling", d["roofline"].get("write_ceiling",{}).get("GB/s"), "of ceiling", d["roofline"].get("frac_of_measured_write_ceiling"), "131072:", d.get("north_star_bank",{}).get("frac_hbm_peak"))
print("   " + " | ".join("%s %.4g ms %.3f%s" % (k, v.get("ms_per_step"), v.get("roofline",{}).get("frac"), ("/%.3f" % v["roofline"]["step_frac"]) if v.get("roofline",{}).get("step_frac") else "") for k,v in d.get("configs",{}).items()))
PY
