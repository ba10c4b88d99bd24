#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullsize.py::test_config4_full_size_fft_mfcc -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for a in "--workload config4" ""; do
echo "== bench.py $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline $a >> $O/bench.log 2>> $O/bench.err
done
tail -c 900 $O/bench.log
