#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && $R/host/dropin_dbg 23010 /tmp/ours.f64 2>&1 | grep -a DBG | tee $O/dbg_ours.txt
