#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for args in "--tune osc_mix_split=1" "--tune osc_mix_split=2" "--mixdown off" "--steps 20 --warmup 5"; do
  echo "== bench.py $args" >> $O/bench.log
  timeout 600 python bench.py --no-cpu-baseline $args >> $O/bench.log 2>> $O/bench.err
done
grep -c value $O/bench.log
