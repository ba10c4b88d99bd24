#!/bin/bash
# round 2, call o: fast_log in the mel stage + exhaustive sqrt candidates probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullparity.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for a in "" "--mfcc-method mfma --mfma-fullk"; do
echo "== bench.py --workload config4 $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline --workload config4 $a >> $O/bench.log 2>> $O/bench.err
done
grep -o '"ms_per_step": [0-9.]*\|"kernels": {[^}]*}[^}]*}' $O/bench.log

timeout 600 python -m pytest tests/test_gpu_convolve.py tests/test_gpu_dropin.py tests/test_gpu_edges.py -m gpu -q -x > $O/pytest2.log 2>&1
tail -2 $O/pytest2.log
