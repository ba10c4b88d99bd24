#!/bin/bash
# round 4: the VALU diet of the oscillator ticks (pair exchange in 4 instructions, 32-bit trip counts, v_fract phase wrap, table copies for
# ds_read_b64) -- parity, then A/B against the build before it and against the single switches, same box, two rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04v
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_osc.py tests/test_gpu_fullparity.py tests/test_gpu_voice.py tests/test_gpu_filter2.py tests/test_gpu_sample.py tests/test_gpu_envgen.py tests/test_gpu_rw_store.py tests/test_gpu_comm.py tests/test_gpu_dropin.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for round in 1 2; do
  for lib in libmaxigpu.so ab_old.so ab_pairasm0.so ab_tab2_0.so; do
    echo "## $lib round $round" | tee -a $O/ab.txt
    MXG_LIB=$R/maximilian_amd/$lib MODE=one REPS=200 timeout 300 python tools/sweep_heavy_osc.py 0 1 9 8 10 2 2>&1 | grep "^wf" | tee -a $O/ab.txt
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
