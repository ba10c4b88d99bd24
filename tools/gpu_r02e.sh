#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_spectral.py tests/test_gpu_fullsize.py::test_config4_full_size_fft_mfcc -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for args in "" "--mixdown off" "--steps 20 --warmup 5" "--workload config4" "--workload config4 --mfcc-method mfma --mfma-fullk"; do
  echo "== bench.py $args" >> $O/bench.log
  timeout 600 python bench.py --no-cpu-baseline $args >> $O/bench.log 2>> $O/bench.err
done
grep -c value $O/bench.log
