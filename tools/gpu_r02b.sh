#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for args in "" "--mixdown off" "--steps 20 --warmup 5" "--workload config4" "--workload config4 --mfcc-method mfma" "--workload config4 --mfcc-method mfma --mfma-fullk"; do
  echo "== bench.py $args" >> $O/bench.log
  timeout 600 python bench.py --no-cpu-baseline $args >> $O/bench.log 2>> $O/bench.err
done
grep -c value $O/bench.log
