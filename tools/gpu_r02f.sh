#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_spectral.py tests/test_gpu_fullsize.py::test_config4_full_size_fft_mfcc -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in 0 1 2 3; do for m in "" "--mix-only"; do
  echo "== bench.py --tune osc_mix_var=$v $m" >> $O/bench.log
  timeout 600 python bench.py --no-cpu-baseline --tune osc_mix_var=$v $m >> $O/bench.log 2>> $O/bench.err
done; done
echo "== bench.py --workload config4" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline --workload config4 >> $O/bench.log 2>> $O/bench.err
grep -c value $O/bench.log
