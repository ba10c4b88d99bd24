#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02af
mkdir -p $O
cd $R


timeout 900 python tools/sweep_grains_general.py 128 > $O/grains_general.txt 2>&1
grep -v amdgpu.ids $O/grains_general.txt
