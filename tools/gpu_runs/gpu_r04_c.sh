#!/bin/bash
# round 4, third GPU call: chunked ticks (all LDS reads of 8 / 16 samples in flight) in K1's pair-row path and in K1m
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py -x -q -m gpu > $O/pytest_osc.log 2>&1
echo "pytest rc=$?" >> $O/pytest_osc.log
tail -3 $O/pytest_osc.log
timeout 600 python tools/sweep_osc_passes.py --voices "" --mix-voices 65536,131072 --out $O/k1m.md > $O/sweep.log 2>&1
timeout 300 python tools/bench_waveforms.py > $O/waveforms.txt 2>&1
for wf in sinebuf4 sawn sinebuf; do
  for sp in 0 1 2; do
    timeout 200 python bench.py --waveform $wf --no-cpu-baseline --no-extras --tune osc_split=$sp --steps 600 > $O/bench_${wf}_split$sp.json 2> /dev/null
  done
done
timeout 300 python bench.py --mixdown fused --no-cpu-baseline > $O/bench_mix.json 2> $O/bench_mix.err
python - <<'PY'
import json, glob, os
O = os.environ.get("O", "gpurun_out/r04c")
for f in sorted(glob.glob("gpurun_out/r04c/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(os.path.basename(f), d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/waveforms.txt | tail -14
sed -n '/65536 voices/,$p' $O/k1m.md | head -50
