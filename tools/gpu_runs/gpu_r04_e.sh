#!/bin/bash
# round 4, fifth GPU call: the software-pipelined K1m (parity + timing), K1 back on sample pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for round in 1 2; do
  for mode in "k1 --no-extras" "k1m --mixdown fused" "k1m_mixonly --mixdown fused --mix-only" "k1m_sawn --mixdown fused --waveform sawn" "k1m_131072 --mixdown fused --voices 131072" "k1m_131072_p2 --mixdown fused --voices 131072 --tune osc_mix_passes=2" "k1m_s2 --mixdown fused --tune osc_mix_split=2"; do
    set -- $mode; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
" | tee -a $O/times.txt
  done
done
timeout 600 python bench.py --gpus 2 --share-gpu --no-cpu-baseline --steps 400 --warmup 50 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks.err
python -c "
import json
d=json.loads([l for l in open('$O/bench_two_ranks_one_gpu.json') if l.startswith('{')][0])
print('two ranks one gpu: step', d['ms_per_step'], 'without reduce', d.get('step_ms_without_reduce'), 'eff', d.get('per_gpu_efficiency'))
"
