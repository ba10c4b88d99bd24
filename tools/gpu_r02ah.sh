#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02aj
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_grains.py tests/test_gpu_extra.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 900 python tools/sweep_grains_general.py 128 > $O/grains_general.txt 2>&1
grep -v amdgpu.ids $O/grains_general.txt
