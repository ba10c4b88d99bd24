#!/bin/bash
# round 4, GPU call: tables kernel (512 threads), the full GPU suite, the default bench line with `configs`
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q -m gpu > $O/pytest_osctab.log 2>&1
echo "pytest rc=$?" >> $O/pytest_osctab.log
tail -4 $O/pytest_osctab.log
timeout 300 python tools/bench_osctab.py 131072,262144 > $O/osctab.txt 2>&1
cat $O/osctab.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04m/bench_default.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["roofline"]["frac"], d.get("north_star_bank",{}).get("frac_hbm_peak"))
for k,v in d.get("configs",{}).items(): print("  ",k, v.get("ms_per_step"), v.get("error"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("step_frac"), v.get("step_vs_headline"))
PY
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -8 $O/pytest_all.log
