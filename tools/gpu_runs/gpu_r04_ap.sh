#!/bin/bash
# round 4: launch shapes for pulse / triangle (0.73 / 0.79 against the ramps' 0.88-0.90)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ap
mkdir -p $O
cd $R
ROUNDS=5 timeout 600 python tools/sweep_heavy_osc.py 6 4 > $O/sweep.txt 2>&1; grep -v amdgpu $O/sweep.txt | awk '/^##/{c=0} {c++; if (c<=9) print}'
