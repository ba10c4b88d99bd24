#!/bin/bash
# round 2, call u: general steady chunk of the ADSR in env_kernel / voice_kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02u
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_voice.py tests/test_gpu_fullparity.py tests/test_gpu_fullsize.py tests/test_gpu_host.py tests/test_gpu_dropin.py tests/test_gpu_edges.py tests/test_gpu_sampler.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python tools/bench_banks.py 2>/dev/null | grep -i "env"
for a in "--workload config3" "--workload config3 --voice-mode 1"; do
timeout 600 python bench.py --no-cpu-baseline $a 2>> $O/bench.err | grep -o '"ms_per_step": [0-9.]*'
done
