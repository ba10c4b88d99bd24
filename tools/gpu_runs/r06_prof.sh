#!/bin/bash
# round 6: rocprofv3 evidence for every workload (kernel-trace + separate PMC passes) -- tools/profile_r06.sh, then tools/summarize_rocprof.py r06
cd $GRAFT_REPO_ROOT
bash tools/profile_r06.sh r06 2>&1 | tail -5
