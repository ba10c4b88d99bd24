#!/bin/bash
# round 4, closing run: the whole GPU suite, smoke(), the default bench line, the per-bank tools -- on the code as committed
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ag
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04ag/bench_default.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["step_ms_gpu"], d["roofline"]["frac"], d.get("north_star_bank",{}).get("frac_hbm_peak"), d["roofline"].get("write_ceiling",{}).get("GB/s"))
for k,v in d.get("configs",{}).items(): print("  ",k, v.get("ms_per_step"), v.get("error"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("step_frac"), v.get("step_vs_headline"))
PY
timeout 300 python tools/bench_banks.py > $O/banks.txt 2>&1
timeout 300 python tools/bench_waveforms.py > $O/waveforms.txt 2>&1
grep -v amdgpu $O/banks.txt; grep -v amdgpu $O/waveforms.txt
