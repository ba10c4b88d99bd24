#!/bin/bash
# round 4: launch shapes of the store-bound oscillators at 65 536 voices (is a full-page store per wavefront -- two voices per lane in two
# time parts -- faster than pair rows?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04af
mkdir -p $O
cd $R
ROUNDS=6 timeout 600 python tools/sweep_heavy_osc.py 8 2 > $O/sweep.txt 2>&1; grep -v amdgpu $O/sweep.txt | awk '/^##/{c=0} {c++; if (c<=12) print}'
