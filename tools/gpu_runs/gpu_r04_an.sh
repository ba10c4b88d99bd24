#!/bin/bash
# round 4: bench.py's fallback exchange (torch tensors + torch.distributed.reduce), forced, two ranks on one GPU over gloo
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04an
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_bench_launch.py -q -x -k "two_ranks" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log | cut -c1-300
