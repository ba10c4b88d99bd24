#!/bin/bash
# round 2, call q: window staging for the interpolating maxiSample players
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_gpu_extra.py tests/test_gpu_edges.py tests/test_gpu_host.py -m gpu -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python tools/bench_banks.py > $O/banks.txt 2> $O/banks.err
cat $O/banks.txt
