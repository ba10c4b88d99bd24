#!/bin/bash
# round 2, call k: full GPU suite after the tolerance / comm changes + the 16-wave fused FFT+MFCC kernel A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for a in "--tune fused_waves16=1" "--tune fused_waves16=0"; do
echo "== bench.py --workload config4 $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline --workload config4 $a >> $O/bench.log 2>> $O/bench.err
done
grep -o '"ms_per_step": [0-9.]*\|"kernels": {[^}]*}[^}]*}' $O/bench.log
