#!/bin/bash
# round 4: regression check of every bench config, new build against the build before the pair-exchange / lean-tick changes, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ab
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_rw_store.py tests/test_gpu_sample.py tests/test_gpu_envgen.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for round in 1 2; do
  for lib in libmaxigpu.so ab_old.so; do
    MXG_LIB=$R/maximilian_amd/$lib timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${lib}_$round.json 2> $O/bench_${lib}_$round.err
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04ab/bench_${lib}_$round.json") if l.startswith("{")][0])
print("$lib round $round headline", d["ms_per_step"], d["step_ms_gpu"], d["roofline"]["frac"], d.get("north_star_bank",{}).get("frac_hbm_peak"))
for k,v in d.get("configs",{}).items(): print("  ",k, v.get("ms_per_step"), v.get("error"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("step_frac"), v.get("step_vs_headline"))
PY
  done
done
