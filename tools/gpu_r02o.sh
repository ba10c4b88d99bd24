#!/bin/bash
# round 2, call o: fast_log in the mel stage + exhaustive sqrt candidates probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullparity.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for a in "" "--mfcc-method mfma --mfma-fullk"; do
echo "== bench.py --workload config4 $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline --workload config4 $a >> $O/bench.log 2>> $O/bench.err
done
grep -o '"ms_per_step": [0-9.]*\|"kernels": {[^}]*}[^}]*}' $O/bench.log
timeout 300 tools/ubench/sqrt_probe > $O/sqrt_probe.txt 2>&1; cat $O/sqrt_probe.txt
