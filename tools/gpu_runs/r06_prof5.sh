#!/bin/bash
# round 6, closing: rocprofv3 passes of the headline and of the workloads the paced schedule changed, on the final build
cd $GRAFT_REPO_ROOT
ONLY="config2 config2_131072 config3 config3_mix config3_modB" bash tools/profile_r06.sh r06 > gpurun_out/prof5.log 2>&1
tail -2 gpurun_out/prof5.log
