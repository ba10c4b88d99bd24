#!/bin/bash
# round 4, seventh GPU call: K1m with its stores spread over the iteration; the launch plan against the knob forms at 393 216+ voices
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_osc.py -x -q -m gpu -k "render_mix" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for round in 1 2; do
  for mode in "k1 --no-extras" "k1m --mixdown fused" "k1m_mixonly --mixdown fused --mix-only" "k1m_sawn --mixdown fused --waveform sawn"; do
    set -- $mode; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/times.txt
  done
done
timeout 900 python tools/sweep_osc_passes.py --voices 393216,524288,1048576 --mix-voices "" --out $O/osc_passes_big.md > $O/sweep.log 2>&1
grep -E "^\*\*|auto|2v sc1 p[1458] " $O/osc_passes_big.md | cut -c1-300
