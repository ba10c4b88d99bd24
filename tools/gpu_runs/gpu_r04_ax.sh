#!/bin/bash
# round 4: the same range for sawn and saw (does the two-pass nt form help them too?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ax
mkdir -p $O
cd $R
WF=10 timeout 600 python tools/sweep_osc_mid.py 73728 81920 > $O/sawn.txt 2>&1; grep -v amdgpu $O/sawn.txt | awk '/^##/{c=0} {c++; if (c<=5 || /^auto/) print}'
WF=3 timeout 600 python tools/sweep_osc_mid.py 73728 81920 > $O/saw.txt 2>&1; grep -v amdgpu $O/saw.txt | awk '/^##/{c=0} {c++; if (c<=5 || /^auto/) print}'
