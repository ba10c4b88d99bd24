#!/bin/bash
# first GPU pass of round 2: tests, then bench lines for every workload / mixdown / event mode
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for args in "" "--kernel-events off" "--kernel-events pass" "--mixdown off" "--mixdown separate" "--steps 20 --warmup 5" \
            "--workload config3" "--workload config3 --voice-mode 1" "--workload config4" "--workload config4 --mfcc-method mfma" "--workload config5" "--workload config5 --mixdown off"; do
  echo "== bench.py $args" >> $O/bench.log
  timeout 600 python bench.py $args >> $O/bench.log 2>> $O/bench.err
done
grep -c value $O/bench.log
