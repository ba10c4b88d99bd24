#!/bin/bash
# round 4: osctab after the table-copy slip (its pipe fetch read the next voice's table: caught by the full suite) -- parity, then the
# two table values of a sample as two ds_read_b64 through an opaque second base against one ds_read2_b64, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ah
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osctab.py tests/test_gpu_osc.py tests/test_bench_launch.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for round in 1 2 3; do
  for lib in libmaxigpu.so ab_tabread2.so; do
    MXG_LIB=$R/maximilian_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 200 --warmup 20 --workload tables 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tables $lib round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'], d['roofline'].get('step_frac'))
" | tee -a $O/ab.txt
  done
done
