#!/bin/bash
# round 4: K1 on the 2 MB-pitch banks -- time parts / passes / numbering sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ac
mkdir -p $O
cd $R
timeout 900 python tools/sweep_osc_pitch.py > $O/pitch.txt 2>&1
grep -v amdgpu $O/pitch.txt | awk '/^##/{c=0} {c++; if (c<=10) print}'
