#!/bin/bash
# round 4: per-waveform tick flavours (lean forms for sinewave / coswave / sinebuf4 / sawn and K1m; K1's sinebuf as it was) -- parity, A/B vs old
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04y
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_osc.py tests/test_gpu_fullparity.py tests/test_gpu_comm.py tests/test_gpu_dropin.py tests/test_gpu_extra.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for round in 1 2 3; do
  for lib in libmaxigpu.so ab_old.so; do
    echo "## $lib round $round" | tee -a $O/ab.txt
    MXG_LIB=$R/maximilian_amd/$lib MODE=one REPS=300 timeout 300 python tools/sweep_heavy_osc.py 8 10 9 0 1 2 4 6 2>&1 | grep "^wf" | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
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
