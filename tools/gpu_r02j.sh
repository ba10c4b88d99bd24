#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_grains.py tests/test_gpu_osc.py tests/test_gpu_fullsize.py::test_config5_full_size_granular_share -m gpu -q -s > $O/pytest.log 2>&1
grep -E "mixdown S=|passed|failed|Error" $O/pytest.log | tail -12
for a in "--workload config5" "--workload config5 --mixdown separate" "--workload config5 --mixdown off"; do
echo "== bench.py $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline $a >> $O/bench.log 2>> $O/bench.err
done
