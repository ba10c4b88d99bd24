#!/bin/bash
# round 4, sixth GPU call: K1m with the VALU-only fold; K1's launch plan (parity + the automatic choice over bank sizes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_osc.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for round in 1 2; do
  for mode in "k1 --no-extras" "k1m --mixdown fused" "k1m_mixonly --mixdown fused --mix-only" "k1m_sawn --mixdown fused --waveform sawn" "k1m_131072_p2 --mixdown fused --voices 131072 --tune osc_mix_passes=2"; do
    set -- $mode; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/times.txt
  done
done
timeout 900 python tools/sweep_osc_auto.py --out $O/osc_auto.md > $O/sweep.log 2>&1
tail -20 $O/osc_auto.md
