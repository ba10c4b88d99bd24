#!/bin/bash
# round 4: the 65 536 ... 98 304-voice rule (sinebuf two-pass nt, sawn pair rows) -- parity, then auto against the knobs again
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ay
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py -q -x -k "plan or passes or config2" 2>&1 | tail -2
timeout 600 python tools/sweep_osc_mid.py 67584 73728 81920 86016 2>&1 | grep -v amdgpu | awk '/^##/{c=0} {c++; if (c<=3 || /^auto/) print}'
WF=10 timeout 600 python tools/sweep_osc_mid.py 81920 2>&1 | grep -v amdgpu | awk '/^##/{c=0} {c++; if (c<=3 || /^auto/) print}'
