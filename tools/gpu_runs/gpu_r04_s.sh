#!/bin/bash
# round 4: the whole GPU suite, smoke(), the default bench line -- on the code as committed
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04s
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04s/bench_default.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["step_ms_gpu"], d["roofline"]["frac"], d.get("north_star_bank",{}).get("frac_hbm_peak"), d["roofline"].get("write_ceiling",{}).get("GB/s"))
for k,v in d.get("configs",{}).items(): print("  ",k, v.get("ms_per_step"), v.get("error"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("step_frac"), v.get("step_vs_headline"))
PY
