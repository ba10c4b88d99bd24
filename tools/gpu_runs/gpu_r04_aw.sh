#!/bin/bash
# round 4: K1 between 65 536 and 98 304 voices (the automatic rule's weak range): shapes x stores x passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04aw
mkdir -p $O
cd $R
timeout 600 python tools/sweep_osc_mid.py 69632 73728 81920 90112 > $O/mid.txt 2>&1; grep -v amdgpu $O/mid.txt
