#!/bin/bash
# round 4, tenth GPU call: where the producer / consumer K1m loses its time (A/B builds: no tile writes, no consumer work, neither)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
for round in 1 2 3; do for lib in libmaxigpu.so ab_pc_noconsume.so ab_pc_notile.so ab_pc_neither.so; do
  for mode in "k1 --no-extras" "k1m_pc --mixdown fused"; do
    set -- $mode; name=$1; shift
    MXG_LIB=$R/maximilian_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', '$lib', 'round $round', 'step_ms', d['ms_per_step'])
" >> $O/ab.txt
  done
done; done
sort $O/ab.txt | awk '{k=$1" "$2; s[k]=s[k]" "$6} END{for(k in s) print k, s[k]}' | sort
