#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02w
mkdir -p $O
cd $R
timeout 900 python tools/sweep_grains_general.py > $O/grains_general.txt 2>&1
cat $O/grains_general.txt | grep -v amdgpu.ids
