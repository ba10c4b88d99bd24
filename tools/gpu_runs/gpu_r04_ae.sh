#!/bin/bash
# round 4: what K1m's producer / consumer form costs over K1 now -- skeleton only / + tile writes / + consumer (A/B builds; timing only,
# the ablated builds do not compute the mix)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ae
mkdir -p $O
cd $R
for round in 1 2 3; do
  for lib in libmaxigpu.so ab_pc_skel.so ab_pc_nocons.so ab_pc_notile.so; do
    MXG_LIB=$R/maximilian_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 600 --warmup 50 --mixdown fused 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('k1m $lib round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))
" | tee -a $O/ab.txt
  done
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-extras --steps 600 --warmup 50 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('k1 round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))
" | tee -a $O/ab.txt
done
