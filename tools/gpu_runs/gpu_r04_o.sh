#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
(cd /tmp && $R/host/dropin_dbg 8010 /tmp/ours.f64 2>&1 | grep DBG; $R/oracle/_ref/example_dbg 8010 /tmp/ref.f64 2>&1 | grep DBG) | tee $O/dbg.txt
timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_osc.py tests/test_gpu_voice.py tests/test_bench_launch.py -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
