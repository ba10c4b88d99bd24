#!/bin/bash
# round 2, call v: steady chunk of maxiEnvGen
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02v
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_envgen.py tests/test_gpu_voice.py tests/test_gpu_edges.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python tools/bench_banks.py 2>/dev/null | grep -i "env"
