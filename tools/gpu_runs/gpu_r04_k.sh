#!/bin/bash
# round 4, eleventh GPU call: the tables extension with 8-voice rounds and a four-deep DMA ring
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 300 python tools/bench_osctab.py > $O/osctab.txt 2>&1
cat $O/osctab.txt
timeout 600 python -m pytest tests/test_gpu_osc.py -x -q -m gpu -k "render_mix" > $O/pytest_mix.log 2>&1
echo "pytest rc=$?" >> $O/pytest_mix.log
tail -3 $O/pytest_mix.log
for round in 1 2 3; do
  for mode in "k1 --no-extras" "k1m_pc --mixdown fused" "k1m_fused --mixdown fused --tune osc_mix_pc=1"; do
    set -- $mode; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'round $round', 'step_ms', d['ms_per_step'])
" | tee -a $O/times.txt
  done
done
