#!/bin/bash
# round 4: K1m's combine window 512 (one barrier per block for the producers) -- parity, then 256 vs 512 interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ad
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_comm.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for round in 1 2 3; do
  for win in 256 512; do
    timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 600 --warmup 50 --mixdown fused --tune osc_mix_pcwin=$win 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('k1m pcwin $win round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/ab.txt
  done
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-extras --steps 600 --warmup 50 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('k1 round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/ab.txt
done
