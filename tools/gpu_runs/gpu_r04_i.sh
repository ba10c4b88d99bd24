#!/bin/bash
# round 4, ninth GPU call: the producer / consumer K1m (parity + timing); the tables extension after pipelining
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_osc.py tests/test_gpu_comm.py tests/test_gpu_osctab.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for round in 1 2; do
  for mode in "k1 --no-extras" "k1m_pc --mixdown fused" "k1m_fused --mixdown fused --tune osc_mix_pc=1" "k1m_pc_mixonly --mixdown fused --mix-only" "k1m_pc_sawn --mixdown fused --waveform sawn" "k1m_pc_saw --mixdown fused --waveform saw" "k1m_pc_131072 --mixdown fused --voices 131072" "k1m_pc_131072_p2 --mixdown fused --voices 131072 --tune osc_mix_passes=2"; do
    set -- $mode; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/times.txt
  done
done
timeout 300 python tools/bench_osctab.py > $O/osctab.txt 2>&1
cat $O/osctab.txt
