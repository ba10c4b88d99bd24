#!/bin/bash
# round 4: config 4's layouts again on the final code (fused_layout 1 / 2, fused_waves16), two rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2; do for t in "fused_layout=0" "fused_layout=1" "fused_layout=2" "fused_waves16=1"; do
  timeout 300 python bench.py --workload config4 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --tune $t 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$t round $round kernel_ms', d['roofline'].get('kernel_ms'), 'step', d['ms_per_step'])
"
done; done
