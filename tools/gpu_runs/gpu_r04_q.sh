#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_r04.sh r04 > gpurun_out/profile_r04.log 2>&1
tail -3 gpurun_out/profile_r04.log
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_osc.py tests/test_gpu_voice.py tests/test_bench_launch.py -q -m gpu > gpurun_out/r04q_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04q_pytest.log
tail -5 gpurun_out/r04q_pytest.log
