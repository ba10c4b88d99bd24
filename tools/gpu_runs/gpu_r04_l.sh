#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python tools/bench_osctab.py 131072,262144 > $O/osctab.txt 2>&1
cat $O/osctab.txt
