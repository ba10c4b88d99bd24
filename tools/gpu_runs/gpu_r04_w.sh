#!/bin/bash
# round 4: which of the tick changes moved K1's store stream (sinebuf 40.6 -> 46 us): single switches against the old build, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04w
mkdir -p $O
cd $R
for round in 1 2; do
  for lib in libmaxigpu.so ab_old.so ab_loop64.so ab_wrapold.so ab_both.so ab_all3.so; do
    echo "## $lib round $round" | tee -a $O/ab.txt
    MXG_LIB=$R/maximilian_amd/$lib MODE=one REPS=200 timeout 300 python tools/sweep_heavy_osc.py 8 10 2 2>&1 | grep "^wf" | tee -a $O/ab.txt
    for mode in "k1 --no-extras" "k1m --mixdown fused"; do
      set -- $mode; name=$1; shift
      MXG_LIB=$R/maximilian_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/ab.txt
    done
  done
done
