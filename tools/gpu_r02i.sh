#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for a in "--waveform sinewave --tune osc_split=1" "--waveform sinewave --tune osc_split=2" "--waveform sinewave --tune osc_split=4" "--out-buffers 8" "--out-buffers 2" "--out-buffers 8 --mixdown fused" "--out-buffers 8 --workload config3"; do
echo "== bench.py $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline $a >> $O/bench.log 2>> $O/bench.err
done
