#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02al
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_dropin.py -m gpu -q -s > $O/pytest.log 2>&1
grep -E "max \|difference|passed|failed|Error|assert" $O/pytest.log | head -40
