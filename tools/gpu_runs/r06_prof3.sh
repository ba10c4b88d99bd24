#!/bin/bash
# round 6, late: the rocprofv3 passes of the three config-3 workloads again (K2f after the small-angle coefficients and the diet)
cd $GRAFT_REPO_ROOT
ONLY="config3 config3_mix config3_modB" bash tools/profile_r06.sh r06 > gpurun_out/prof3.log 2>&1
tail -3 gpurun_out/prof3.log
